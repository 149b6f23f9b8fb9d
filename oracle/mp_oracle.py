"""CPU oracle for the MobilePoser per-frame inference path (numpy, float32).

TEST INFRASTRUCTURE ONLY.  Nothing under ``mobileposer_amd/`` imports this file; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may.  It is a restatement,
written from the reference's behaviour, of the functions listed in SURVEY.md section 8(a); each
function cites the reference lines it follows (paths relative to /root/reference/mobileposer).

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference itself in the build
container and records inputs/outputs under ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function here against those vectors.  Not pinned (module absent from the reference,
SURVEY.md F4): ``dynamics.PhysicsOptimizer`` -- not restated.

Third-party arithmetic restated from its published definition: ``torch.nn.LSTM`` / ``nn.Linear``
(torch 2.1.2 pinned by the reference's requirements.txt:121): gates i,f,g,o;
c' = sigmoid(f)*c + sigmoid(i)*tanh(g); h' = sigmoid(o)*tanh(c').
"""
import numpy as np

F32 = np.float32

REDUCED = [0, 1, 2, 3, 4, 5, 6, 9, 12, 13, 14, 15, 16, 17, 18, 19]     # config.py:134
IGNORED = [0, 7, 8, 10, 11, 20, 21, 22, 23]                             # config.py:135
PARENT = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
GRAVITY_VELOCITY = -0.018                                               # config.py:131
VEL_DIVISOR = 30 / 2                                                    # fps / vel_scale, net.py:141
PAST, FUTURE = 40, 5                                                    # config.py:52-53

PREFIX = {"joints": "joints.joints.", "pose": "pose.pose.",
          "foot_contact": "foot_contact.footcontact.", "velocity": "velocity.vel."}


def _sigmoid(x):
    return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


# --------------------------------------------------------------------------------------------
# a1: RNN.forward (models/rnn.py:20-33) = Linear -> ReLU -> (dropout: identity in eval) ->
#     pack_padded_sequence -> nn.LSTM(2 layers, bi/uni) -> pad_packed_sequence -> Linear
# --------------------------------------------------------------------------------------------
def _lstm_direction(xs, lengths, w_ih, w_hh, b_ih, b_hh, h0, c0, reverse):
    """One layer, one direction, packed-sequence semantics.

    xs [B,T,in]; lengths [B]; returns out [B,T,H] (zeros at t >= len_b), h_n, c_n [B,H].
    Forward walks t = 0..len_b-1; reverse walks t = len_b-1..0 *per sequence* starting from the
    initial state (rnn.py:25 pack -> the reverse direction starts at each sequence's own last
    valid frame).  h_n/c_n are the state after the last step each sequence actually took.
    """
    B, T, _ = xs.shape
    H = w_hh.shape[1]
    h = h0.astype(F32).copy()
    c = c0.astype(F32).copy()
    out = np.zeros((B, T, H), dtype=F32)
    bias = (b_ih + b_hh).astype(F32)
    xproj = (xs.reshape(B * T, -1) @ w_ih.T).reshape(B, T, 4 * H).astype(F32)
    w_hh_t = np.ascontiguousarray(w_hh.T)
    rows = np.arange(B)
    for s in range(T):
        active = lengths > s
        if not active.any():
            break
        t_idx = np.where(active, (lengths - 1 - s) if reverse else s, 0)
        g = xproj[rows, t_idx] + h @ w_hh_t + bias
        i = _sigmoid(g[:, 0 * H:1 * H])
        f = _sigmoid(g[:, 1 * H:2 * H])
        gg = np.tanh(g[:, 2 * H:3 * H], dtype=F32)
        o = _sigmoid(g[:, 3 * H:4 * H])
        c_new = (f * c + i * gg).astype(F32)
        h_new = (o * np.tanh(c_new, dtype=F32)).astype(F32)
        a = active[:, None]
        c = np.where(a, c_new, c)
        h = np.where(a, h_new, h)
        out[rows[active], t_idx[active]] = h_new[active]
    return out, h, c


def rnn_forward(sd, prefix, x, lengths, state=None):
    """RNN.forward (models/rnn.py:20-33).

    sd: state dict (key -> ndarray); prefix e.g. 'joints.joints.'; x [B,T,n_in]; lengths list[int] or None (time-major, below).
    state: None or (h0, c0) each [layers*dirs, B, H] in nn.LSTM order (l0, l0_reverse, l1, l1_reverse).
    Returns y [B, max(lengths), n_out], (h_n, c_n).
    """
    x = np.asarray(x, dtype=F32)
    if lengths is None:
        # rnn.py:23-28: without seq_lengths nothing is packed and nn.LSTM -- built without batch_first (rnn.py:15) -- reads dim 0
        # of [B,T,n] as TIME and dim 1 as the batch: T sequences of B steps, a state of batch T; the linear layers
        # work row by row, so the output keeps the caller's layout
        B, T, _ = x.shape
        y, st = rnn_forward(sd, prefix, np.ascontiguousarray(x.transpose(1, 0, 2)), [B] * T, state)
        return np.ascontiguousarray(y.transpose(1, 0, 2)), st
    lengths = np.asarray(lengths, dtype=np.int64)
    B, T, _ = x.shape
    w1, b1 = sd[prefix + "linear1.weight"], sd[prefix + "linear1.bias"]
    data = np.maximum(x.reshape(B * T, -1) @ w1.T + b1, F32(0)).astype(F32).reshape(B, T, -1)   # rnn.py:22
    H = w1.shape[0]
    bidir = (prefix + "rnn.weight_ih_l0_reverse") in sd
    dirs = 2 if bidir else 1
    h_n, c_n = [], []
    for layer in range(2):
        outs = []
        for d in range(dirs):
            sfx = f"_l{layer}" + ("_reverse" if d == 1 else "")
            k = layer * dirs + d
            h0 = state[0][k] if state is not None else np.zeros((B, H), dtype=F32)
            c0 = state[1][k] if state is not None else np.zeros((B, H), dtype=F32)
            out, h, c = _lstm_direction(
                data, lengths, sd[prefix + "rnn.weight_ih" + sfx], sd[prefix + "rnn.weight_hh" + sfx],
                sd[prefix + "rnn.bias_ih" + sfx], sd[prefix + "rnn.bias_hh" + sfx], h0, c0, reverse=(d == 1))
            outs.append(out)
            h_n.append(h)
            c_n.append(c)
        data = np.concatenate(outs, axis=-1)
    t_max = int(lengths.max())                                            # pad_packed_sequence, rnn.py:31
    data = data[:, :t_max]
    w2, b2 = sd[prefix + "linear2.weight"], sd[prefix + "linear2.bias"]
    y = (data.reshape(B * t_max, -1) @ w2.T + b2).astype(F32).reshape(B, t_max, -1)             # rnn.py:32
    return y, (np.stack(h_n), np.stack(c_n))


# --------------------------------------------------------------------------------------------
# a7: r6d -> R, reduced -> full, global -> local
# --------------------------------------------------------------------------------------------
def r6d_to_rotation_matrix(r6d):
    """articulate/math/angular.py:167-182 (+ normalize_tensor, general.py:27-39).

    The 6 numbers are the first two COLUMNS of R (stack(dim=-1), angular.py:180); NaN -> 0 (:181).
    """
    r = np.asarray(r6d, dtype=F32).reshape(-1, 6)
    a, b = r[:, 0:3], r[:, 3:6]
    with np.errstate(divide="ignore", invalid="ignore"):
        c0 = (a / np.sqrt((a * a).sum(axis=1, keepdims=True), dtype=F32)).astype(F32)
        u = (b - (c0 * b).sum(axis=1, keepdims=True) * c0).astype(F32)
        c1 = (u / np.sqrt((u * u).sum(axis=1, keepdims=True), dtype=F32)).astype(F32)
        c2 = np.cross(c0, c1).astype(F32)
    R = np.stack((c0, c1, c2), axis=-1)
    R[np.isnan(R)] = 0
    return R.astype(F32)


def reduced_pose_to_full(reduced):
    """utils/model_utils.py:18-25: scatter 16 -> 24 joints, identity elsewhere.  reduced [N,16,3,3]."""
    N = reduced.shape[0]
    full = np.tile(np.eye(3, dtype=F32), (N, 24, 1, 1))
    full[:, REDUCED] = reduced
    return full


def inverse_kinematics_R(R_global, parent=PARENT):
    """articulate/math/spatial.py:197-221 via _inverse_tree (:115-123): R_loc[i] = R_glb[p(i)]^T R_glb[i]."""
    R_global = np.asarray(R_global, dtype=F32).reshape(-1, 24, 3, 3)
    loc = np.empty_like(R_global)
    loc[:, 0] = R_global[:, 0]
    for i in range(1, 24):
        loc[:, i] = np.matmul(R_global[:, parent[i]].transpose(0, 2, 1), R_global[:, i])
    return loc


def reduced_global_to_full(r6d_96, parent=PARENT):
    """MobilePoserNet._reduced_global_to_full (models/net.py:93-99).  r6d_96 [..., 96] -> [N,24,3,3]."""
    R = r6d_to_rotation_matrix(r6d_96).reshape(-1, 16, 3, 3)
    glb = reduced_pose_to_full(R)
    loc = inverse_kinematics_R(glb, parent)
    loc[:, IGNORED] = np.eye(3, dtype=F32)
    loc[:, 0] = glb[:, 0]
    return loc


# --------------------------------------------------------------------------------------------
# a12: forward kinematics (no mesh)
# --------------------------------------------------------------------------------------------
def forward_kinematics(pose, J, parent=PARENT, tran=None):
    """ParametricModel.forward_kinematics, calc_mesh=False (articulate/model.py:208-232).

    pose [N,24,3,3] local rotations; J [24,3] raw SMPL joints (root-aligned inside, model.py:87).
    Returns R_global [N,24,3,3], joint [N,24,3].  G_i = G_p(i) * [R_i b_i; 0 1], b_i = j_i - j_p(i)
    (spatial.py:60-75,148-167,224-249,104-112).
    """
    pose = np.asarray(pose, dtype=F32).reshape(-1, 24, 3, 3)
    j = (J - J[:1]).astype(F32)
    bone = j.copy()
    for i in range(1, 24):
        bone[i] = j[i] - j[parent[i]]
    N = pose.shape[0]
    Rg = np.empty((N, 24, 3, 3), dtype=F32)
    pg = np.empty((N, 24, 3), dtype=F32)
    Rg[:, 0] = pose[:, 0]
    pg[:, 0] = bone[0]
    for i in range(1, 24):
        p = parent[i]
        Rg[:, i] = np.matmul(Rg[:, p], pose[:, i])
        pg[:, i] = (np.matmul(Rg[:, p], bone[i][None, :, None])[..., 0] + pg[:, p]).astype(F32)
    if tran is not None:
        pg = pg + np.asarray(tran, dtype=F32).reshape(-1, 1, 3)
    return Rg, pg


def forward_kinematics_mesh(pose, smpl, parent=PARENT, tran=None):
    """ParametricModel.forward_kinematics with calc_mesh=True, shape=None, no pose blendshape
    (articulate/model.py:208-240).  smpl: dict with 'J', 'v_template', 'weights'."""
    pose = np.asarray(pose, dtype=F32).reshape(-1, 24, 3, 3)
    J = np.asarray(smpl["J"], dtype=F32)
    Rg, pg = forward_kinematics(pose, J, parent)
    j = (J - J[:1]).astype(F32)
    v = (np.asarray(smpl["v_template"], dtype=F32) - J[:1]).astype(F32)
    # T_global[..., -1:] -= T_global @ [j;0]  (model.py:234): translation becomes p_g - R_g j
    tg = pg - np.einsum("njab,jb->nja", Rg, j)
    W = np.asarray(smpl["weights"], dtype=F32)                 # [V,24]
    Rv = np.einsum("njab,vj->nvab", Rg, W)
    tv = np.einsum("nja,vj->nva", tg, W)
    vert = (np.einsum("nvab,vb->nva", Rv, v) + tv).astype(F32)
    if tran is not None:
        t = np.asarray(tran, dtype=F32).reshape(-1, 1, 3)
        pg, vert = pg + t, vert + t
    return Rg, pg, vert


def shaped_body(smpl, shape):
    """ParametricModel.get_zero_pose_joint_and_vertex(shape) (articulate/model.py:84-89):
    v = shapedirs . shape + v_template;  j = J_regressor v;  j, v = j - j[0], v - j[0].
    shape [S,10] -> (j [S,24,3], v [S,V,3]), both root-aligned."""
    shape = np.asarray(shape, dtype=F32).reshape(-1, 10)
    sd = np.asarray(smpl["shapedirs"], dtype=F32)                              # [V,3,10]
    jreg = smpl["J_regressor"]
    jreg = np.asarray(jreg.toarray() if hasattr(jreg, "toarray") else jreg, dtype=F32)   # model.py:29
    v = (np.einsum("sk,vck->svc", shape, sd) + np.asarray(smpl["v_template"], dtype=F32)).astype(F32)
    j = np.einsum("jv,svc->sjc", jreg, v).astype(F32)
    return (j - j[:, :1]).astype(F32), (v - j[:, :1]).astype(F32)


def forward_kinematics_shape(pose, smpl, shape, parent=PARENT, tran=None, pose_blendshape=False):
    """ParametricModel.forward_kinematics with shape != None, calc_mesh=True (articulate/model.py:208-240).
    shape None (mean shape) | [10] | [1,10] | [N,10].  ``pose_blendshape``: v += posedirs . (pose[1:] - I) (model.py:236-238,
    use_pose_blendshape=True).  Returns R_global [N,24,3,3], joint [N,24,3], vert [N,V,3]."""
    pose = np.asarray(pose, dtype=F32).reshape(-1, 24, 3, 3)
    N = pose.shape[0]
    if shape is None:                                                          # model.py:82
        J = np.asarray(smpl["J"], dtype=F32)
        j, v = (J - J[:1])[None], (np.asarray(smpl["v_template"], dtype=F32) - J[:1])[None]
    else:
        j, v = shaped_body(smpl, shape)
    j = np.broadcast_to(j, (N, 24, 3))
    v = np.broadcast_to(v, (N,) + v.shape[1:])
    if pose_blendshape:
        r = (pose[:, 1:] - np.eye(3, dtype=F32)).reshape(N, 207)
        v = (v + np.einsum("nk,vck->nvc", r, np.asarray(smpl["posedirs"], dtype=F32))).astype(F32)
    bone = j.copy()
    for i in range(1, 24):
        bone[:, i] = j[:, i] - j[:, parent[i]]
    Rg = np.empty((N, 24, 3, 3), dtype=F32)
    pg = np.empty((N, 24, 3), dtype=F32)
    Rg[:, 0] = pose[:, 0]
    pg[:, 0] = bone[:, 0]
    for i in range(1, 24):
        p = parent[i]
        Rg[:, i] = np.matmul(Rg[:, p], pose[:, i])
        pg[:, i] = (np.einsum("nab,nb->na", Rg[:, p], bone[:, i]) + pg[:, p]).astype(F32)
    tg = pg - np.einsum("njab,njb->nja", Rg, j)
    W = np.asarray(smpl["weights"], dtype=F32)
    Rv = np.einsum("njab,vj->nvab", Rg, W)
    tv = np.einsum("nja,vj->nva", tg, W)
    vert = (np.einsum("nvab,nvb->nva", Rv, v) + tv).astype(F32)
    if tran is not None:
        t = np.asarray(tran, dtype=F32).reshape(-1, 1, 3)
        pg, vert = pg + t, vert + t
    return Rg, pg, vert


# --------------------------------------------------------------------------------------------
# a6/a8/a9: the orchestrator with its state (models/net.py)
# --------------------------------------------------------------------------------------------
def _prob_to_weight(p):
    """net.py:90-91."""
    lo, hi = 0.5, 0.9
    return (np.clip(p, lo, hi) - lo) / (hi - lo)


class OracleNet:
    """State machine mirroring MobilePoserNet (models/net.py:22-219) incl. quirks Q1-Q8 of SURVEY 8(a)."""

    def __init__(self, sd, J, parent=PARENT):
        self.sd = {k: np.asarray(v, dtype=F32) for k, v in sd.items()}
        self.parent = list(parent)
        self.J = np.asarray(J, dtype=F32)
        self.j = (self.J - self.J[:1]).astype(F32)                        # model.py:87
        self.feet_pos = self.j[10:12].copy()                              # net.py:48
        self.floor_y = float(self.j[10:12, 1].min())                      # net.py:49
        self.gravity_velocity = np.array([0, GRAVITY_VELOCITY, 0], dtype=F32)
        self.last_lfoot_pos, self.last_rfoot_pos = self.feet_pos[0].copy(), self.feet_pos[1].copy()   # net.py:59
        self.velocity_rnn_state = None                                    # velocity.py:30 (Q1: survives reset())
        self.reset()

    def reset(self):
        """net.py:84-88 -- note: does NOT clear the velocity module's rnn_state, nor last foot positions."""
        self.imu = None
        self.current_root_y = 0
        self.last_root_pos = np.zeros(3, dtype=F32)

    def forward(self, batch, input_lengths):
        """net.py:101-119.  input_lengths=None: nn.LSTM is built without batch_first (rnn.py:15) and only the packed
        path is batch-first (rnn.py:25), so dim 0 of [B,T,60] is TIME and dim 1 the batch -- T sequences of B steps, the carried
        velocity state of batch T; the linear layers and net.py:110 work row by row, so the outputs keep the caller's layout."""
        batch = np.asarray(batch, dtype=F32)
        if input_lengths is None:
            B, T = batch.shape[0], batch.shape[1]
            pose, joints, vel, contact = self.forward(np.ascontiguousarray(batch.transpose(1, 0, 2)), [B] * T)
            back = lambda a: np.ascontiguousarray(np.asarray(a).reshape(T, B, -1).transpose(1, 0, 2))
            self._last_r6d, self._last_vel = back(self._last_r6d), back(self._last_vel)
            pose = np.ascontiguousarray(np.asarray(pose).reshape(T, B, 24, 3, 3).transpose(1, 0, 2, 3, 4)).reshape(B * T, 24, 3, 3)
            return pose, back(joints), back(vel), back(contact)
        joints, _ = rnn_forward(self.sd, PREFIX["joints"], batch, input_lengths)                  # :103
        t_max = joints.shape[1]
        x132 = np.concatenate((joints, batch[:, :t_max]), axis=-1)                                  # :106
        r6d, _ = rnn_forward(self.sd, PREFIX["pose"], x132, input_lengths)                        # :107
        pose = reduced_global_to_full(r6d, self.parent)                                            # :110
        contact, _ = rnn_forward(self.sd, PREFIX["foot_contact"], x132, input_lengths)            # :114
        vel, self.velocity_rnn_state = rnn_forward(self.sd, PREFIX["velocity"], x132, input_lengths,
                                                   self.velocity_rnn_state)                        # :117, velocity.py:45-48
        self._last_r6d, self._last_vel = r6d, vel
        return pose, joints, vel, contact

    def forward_offline(self, imu, input_lengths):
        """net.py:121-171 (PHYSICS off).  imu [1,T,60] -> pose [T,24,3,3], joints [1,T,72], tran [T,3], contact [T,2]."""
        pose, pred_joints, vel, contact = self.forward(imu, input_lengths)
        contact = contact[0]
        vel = vel[0]                                                       # .squeeze(0), net.py:117
        joints = pred_joints[0].reshape(-1, 24, 3)
        tran = translate_offline(joints, vel, contact, self.floor_y)
        return pose, pred_joints, tran, contact

    def forward_online(self, data):
        """net.py:173-219 (PHYSICS off).  data [60]."""
        data = np.asarray(data, dtype=F32).reshape(-1)
        total = PAST + FUTURE
        imu = np.tile(data, (total, 1)) if self.imu is None else np.concatenate((self.imu[1:], data[None]))   # :175
        pose, pred_joints, vel, contact = self.forward(imu[None], [total])                          # :178
        pose = pose[PAST].reshape(-1, 9)                                                             # :181
        joints = pred_joints[0][PAST].reshape(24, 3)                                                 # :184
        contact = contact[0][PAST]                                                                   # :187
        lfoot, rfoot = joints[10], joints[11]
        if contact[0] > contact[1]:                                                                  # :189
            contact_vel = self.last_lfoot_pos - lfoot + self.gravity_velocity
        else:
            contact_vel = self.last_rfoot_pos - rfoot + self.gravity_velocity
        root_vel = vel[0].reshape(-1, 24, 3)[:, 0]
        pred_vel = (root_vel[PAST] / F32(VEL_DIVISOR)).astype(F32)                                   # :196
        weight = F32(_prob_to_weight(contact.max()))                                                 # :197 (raw logit, Q5)
        velocity = (pred_vel * (F32(1) - weight) + contact_vel * weight).astype(F32)                 # :198, general.py:24
        current_foot_y = self.current_root_y + min(float(lfoot[1]), float(rfoot[1]))                 # :201
        if current_foot_y + float(velocity[1]) <= self.floor_y:                                      # :202
            velocity[1] = self.floor_y - current_foot_y
        self.current_root_y += float(velocity[1])                                                    # :205
        self.last_lfoot_pos, self.last_rfoot_pos = lfoot.copy(), rfoot.copy()
        self.imu = imu
        self.last_root_pos = (self.last_root_pos + velocity).astype(F32)                             # :208
        return pose, pred_joints[0], self.last_root_pos.copy(), contact


def translate_offline(joints, vel, contact, floor_y):
    """The translation solver inside forward_offline (net.py:130-154), one sequence.

    joints [T,24,3], vel [T,72], contact [T,2] raw logits -> tran [T,3].
    """
    joints = np.asarray(joints, dtype=F32).reshape(-1, 24, 3)
    T = joints.shape[0]
    g = np.array([0, GRAVITY_VELOCITY, 0], dtype=F32)
    zeros = np.zeros((1, 3), dtype=F32)
    dl = np.concatenate((zeros, joints[:-1, 10] - joints[1:, 10]))                                   # :134
    dr = np.concatenate((zeros, joints[:-1, 11] - joints[1:, 11]))                                   # :135
    idx = np.argmax(contact, axis=1).reshape(-1, 1)                                                  # :136 (ties -> 0)
    # lerp with an integer tensor t: a*(1-t)+b*t (general.py:24)
    contact_vel = (g + (dl * (1 - idx) + dr * idx)).astype(F32)
    pred_vel = (vel.reshape(-1, 24, 3)[:, 0] / F32(VEL_DIVISOR)).astype(F32)                          # :140-141
    weight = _prob_to_weight(_sigmoid(contact.max(axis=1).astype(F32))).astype(F32).reshape(-1, 1)    # :144
    velocity = (pred_vel * (F32(1) - weight) + contact_vel * weight).astype(F32)                      # :145
    current_root_y = 0                                                                                # :148
    for i in range(T):                                                                                # :149-153
        current_foot_y = current_root_y + float(joints[i, 10:12, 1].min())
        if current_foot_y + float(velocity[i, 1]) <= floor_y:
            velocity[i, 1] = floor_y - current_foot_y
        current_root_y += float(velocity[i, 1])
    # tran[i] = velocity[:i+1].sum(0) (:154): an fp32 re-summation of the clamped velocities (the O(T^2) form up to T = 512,
    # a float64 running sum beyond; the long branch is pinned by golden G14's T = 600 sequence: 4e-6 from the reference)
    tran = np.stack([velocity[:i + 1].sum(axis=0, dtype=F32) for i in range(T)]).astype(F32) if T <= 512 \
        else np.cumsum(velocity.astype(np.float64), axis=0).astype(F32)
    return tran


# --------------------------------------------------------------------------------------------
# a15: FullMotionEvaluator.__call__ (articulate/evaluator.py:292-343)
# --------------------------------------------------------------------------------------------
def angle_between(Ra, Rb):
    """articulate/math/angular.py:86-99: |rotvec(Ra^T Rb)| in radians (the reference goes through cv2.Rodrigues)."""
    D = np.matmul(np.swapaxes(np.asarray(Ra, dtype=np.float64), -1, -2), np.asarray(Rb, dtype=np.float64))
    n = np.linalg.norm(D - np.eye(3), axis=(-1, -2))
    return 2.0 * np.arcsin(np.clip(n / (2.0 * np.sqrt(2.0)), 0.0, 1.0))


def full_motion_evaluator(pose_p, pose_t, smpl, tran_p=None, tran_t=None, fps=30, joint_mask=(2, 5, 16, 20),
                          align_joint=0, parent=PARENT):
    """The 10 x [mean, std] error table of FullMotionEvaluator.__call__ (evaluator.py:292-343), mean shape.
    std follows torch: ``x.std(dim=0).mean()`` with Bessel's correction."""
    f = fps
    pose_p = np.asarray(pose_p, dtype=F32).reshape(-1, 24, 3, 3)
    pose_t = np.asarray(pose_t, dtype=F32).reshape(-1, 24, 3, 3)
    Rg_p, j_p, v_p = forward_kinematics_mesh(pose_p, smpl, parent, tran_p)
    Rg_t, j_t, v_t = forward_kinematics_mesh(pose_t, smpl, parent, tran_t)
    off = (j_t[:, align_joint] - j_p[:, align_joint])[:, None]
    ve = np.linalg.norm(v_p + off - v_t, axis=2)
    je = np.linalg.norm(j_p + off - j_t, axis=2)
    lae = np.degrees(angle_between(pose_p, pose_t))
    gae = np.degrees(angle_between(Rg_p, Rg_t))
    jkp = np.linalg.norm((j_p[3:] - 3 * j_p[2:-1] + 3 * j_p[1:-2] - j_p[:-3]) * (f ** 3), axis=2)
    jkt = np.linalg.norm((j_t[3:] - 3 * j_t[2:-1] + 3 * j_t[1:-2] - j_t[:-3]) * (f ** 3), axis=2)
    te = np.linalg.norm((j_p[f:, :1] - j_p[:-f, :1]) - (j_t[f:, :1] - j_t[:-f, :1]), axis=2) * 100
    m = list(joint_mask)
    rows = [je, ve, lae, gae, jkp, jkt, te, je[:, m], lae[:, m], gae[:, m]]
    return np.array([[x.mean(), x.std(axis=0, ddof=1).mean()] for x in rows], dtype=np.float64)

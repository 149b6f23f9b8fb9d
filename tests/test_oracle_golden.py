"""Pin the CPU oracle (oracle/mp_oracle.py) against golden vectors recorded from the reference itself
(tests/golden/make_golden.py).  CPU only.  Tolerances: 1e-4 on joint angles / raw outputs, 1 mm on
root translation (BASELINE.json north_star); the oracle actually agrees to ~1e-6."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, geodesic, load_golden
from oracle import mp_oracle as O
from mobileposer_amd.manifest import state_dict_manifest

TOL = 1e-4
TIGHT = 2e-5


def test_g7_manifest_matches_reference():
    with open(os.path.join(GOLDEN, "g7_manifest.json")) as f:
        g = json.load(f)
    ours = [[k, list(s)] for k, s in state_dict_manifest().items()]
    assert ours == g["keys"]
    assert g["parent"] == O.PARENT


def test_g7_floor_and_feet(weights, smpl):
    with open(os.path.join(GOLDEN, "g7_manifest.json")) as f:
        g = json.load(f)
    net = O.OracleNet(weights, smpl["J"])
    assert abs(net.floor_y - g["floor_y"]) < 1e-7
    np.testing.assert_allclose(net.feet_pos, np.array(g["feet_pos"], dtype=np.float32), atol=1e-7)


@pytest.mark.parametrize("name", ["joints", "pose", "foot_contact", "velocity"])
def test_g1_rnn_ragged(weights, name):
    g = load_golden("g1_rnn.npz")
    lengths = g["lengths"].tolist()
    y, (h, c) = O.rnn_forward(weights, O.PREFIX[name], g[f"{name}_x"], lengths)
    assert y.shape == g[f"{name}_y"].shape
    assert np.abs(y - g[f"{name}_y"]).max() < TIGHT
    assert np.abs(h - g[f"{name}_h"]).max() < TIGHT
    assert np.abs(c - g[f"{name}_c"]).max() < TIGHT
    # carried-in state (velocity.py:47)
    y2, (h2, c2) = O.rnn_forward(weights, O.PREFIX[name], g[f"{name}_x"], lengths, (g[f"{name}_h"], g[f"{name}_c"]))
    assert np.abs(y2 - g[f"{name}_y2"]).max() < TIGHT
    assert np.abs(h2 - g[f"{name}_h2"]).max() < TIGHT
    assert np.abs(c2 - g[f"{name}_c2"]).max() < TIGHT
    # Q4: padded positions carry linear2.bias exactly
    b2 = weights[O.PREFIX[name] + "linear2.bias"]
    np.testing.assert_allclose(y[1, 9:], np.broadcast_to(b2, y[1, 9:].shape), atol=1e-7)


@pytest.mark.parametrize("tag", ["eq", "rag"])
def test_g2_forward(weights, smpl, tag):
    g = load_golden("g2_forward.npz")
    net = O.OracleNet(weights, smpl["J"])
    pose, joints, vel, contact = net.forward(g["imu"], g[f"{tag}_lengths"].tolist())
    assert pose.shape == g[f"{tag}_pose"].shape           # [B*T,24,3,3] (Q8 flattening)
    assert np.abs(net._last_r6d - g[f"{tag}_r6d"]).max() < TIGHT
    assert np.abs(joints - g[f"{tag}_joints"]).max() < TIGHT
    assert np.abs(vel - g[f"{tag}_vel"]).max() < TIGHT
    assert np.abs(contact - g[f"{tag}_contact"]).max() < TIGHT
    assert geodesic(pose, g[f"{tag}_pose"]).max() < TOL
    assert np.abs(pose - g[f"{tag}_pose"]).max() < TOL
    h, c = net.velocity_rnn_state
    assert np.abs(h - g[f"{tag}_vel_h"]).max() < TIGHT and np.abs(c - g[f"{tag}_vel_c"]).max() < TIGHT


def test_g3_r6d_ik_degenerate():
    g = load_golden("g3_r6d_ik.npz")
    rot = O.r6d_to_rotation_matrix(g["r6d"])
    assert not np.isnan(rot).any()
    assert np.abs(rot - g["rot"]).max() < TIGHT
    pose = O.reduced_global_to_full(g["r6d"])
    assert np.abs(pose - g["pose"]).max() < TIGHT
    # the degenerate rows really are degenerate in the golden (NaN -> 0 path exercised)
    assert np.abs(g["rot"].reshape(64, 16, 3, 3)[5, 0]).max() == 0.0


def test_g4_offline_stale_velocity_state(weights, smpl):
    g = load_golden("g4_offline.npz")
    net = O.OracleNet(weights, smpl["J"])
    res = {}
    for tag, x in (("a", g["imu_a"]), ("b", g["imu_b"]), ("a_again", g["imu_a"])):
        net.reset()
        pose, joints, tran, contact = net.forward_offline(x, [x.shape[1]])
        res[tag] = tran
        assert geodesic(pose, g[f"{tag}_pose"]).max() < TOL
        assert np.abs(joints - g[f"{tag}_joints"]).max() < TIGHT
        assert np.abs(contact - g[f"{tag}_contact"]).max() < TIGHT
        assert np.abs(tran - g[f"{tag}_tran"]).max() < 1e-3, tag        # 1 mm
        assert np.abs(tran - g[f"{tag}_tran"]).max() < 5e-5, tag
    # Q1: the same input gives a different translation the second time ...
    assert np.abs(g["a_again_tran"] - g["a_tran"]).max() > 1e-5
    # ... and clearing the velocity state restores it
    net.reset()
    net.velocity_rnn_state = None
    _, _, tran, _ = net.forward_offline(g["imu_a"], [g["imu_a"].shape[1]])
    assert np.abs(tran - g["a_cleared_tran"]).max() < 5e-5
    assert np.abs(g["a_cleared_tran"] - g["a_tran"]).max() < 1e-6


def test_g4_floor_clamp_is_exercised(weights, smpl):
    """The golden sequence must actually hit the floor-penetration branch (net.py:151-152)."""
    g = load_golden("g4_offline.npz")
    net = O.OracleNet(weights, smpl["J"])
    tran = g["a_tran"]
    joints = g["a_joints"][0].reshape(-1, 24, 3)
    foot_y = tran[:, 1] + joints[:, 10:12, 1].min(axis=1)
    on_floor = np.abs(foot_y - net.floor_y) < 1e-4
    assert on_floor.sum() > 10 and (~on_floor).sum() > 10


def test_g5_online(weights, smpl):
    g = load_golden("g5_online.npz")
    net = O.OracleNet(weights, smpl["J"])
    net.reset()
    for k, f in enumerate(g["imu"]):
        pose, joints, tran, contact = net.forward_online(f)
        assert geodesic(pose.reshape(24, 3, 3), g["pose"][k].reshape(24, 3, 3)).max() < TOL
        assert np.abs(joints[40] - g["joints40"][k]).max() < TIGHT
        assert np.abs(contact - g["contact"][k]).max() < TIGHT
        assert np.abs(tran - g["tran"][k]).max() < 1e-4, k
    h, c = net.velocity_rnn_state
    assert np.abs(h - g["vel_h"]).max() < TIGHT and np.abs(c - g["vel_c"]).max() < TIGHT
    assert abs(net.current_root_y - float(g["current_root_y"])) < 1e-4


def test_g6_fk(smpl):
    g = load_golden("g6_fk.npz")
    Rg, jg = O.forward_kinematics(g["pose"], smpl["J"])
    assert np.abs(Rg - g["R_global"]).max() < TIGHT
    assert np.abs(jg - g["joint"]).max() < TIGHT
    Rg2, jg2, vg2 = O.forward_kinematics_mesh(g["pose"], smpl, tran=g["tran"])
    assert np.abs(jg2 - g["joint_tran"]).max() < TIGHT
    assert np.abs(vg2 - g["vert_tran"]).max() < TIGHT


def test_g12_fk_with_shape(smpl):
    """forward_kinematics(pose, shape, tran, calc_mesh=True) -- one shared shape and a shape per frame."""
    g = load_golden("g12_fk_shape.npz")
    for tag in ("one", "per"):
        Rg, jg, vg = O.forward_kinematics_shape(g["pose"], smpl, g[f"{tag}_shape"], tran=g["tran"])
        assert np.abs(Rg - g[f"{tag}_R"]).max() < TIGHT
        assert np.abs(jg - g[f"{tag}_joint"]).max() < TIGHT
        assert np.abs(vg - g[f"{tag}_vert"]).max() < TIGHT
        _, jg2, _ = O.forward_kinematics_shape(g["pose"], smpl, g[f"{tag}_shape"])
        assert np.abs(jg2 - g[f"{tag}_joint_notran"]).max() < TIGHT


@pytest.mark.parametrize("tag", ["eq", "rag"])
def test_torch_baseline_restatement_matches_golden(weights, smpl, tag):
    """oracle/torch_ref.py (the torch-CPU leg of bench.py's cpu_baseline) against the reference's own outputs."""
    from oracle.torch_ref import TorchNet
    g = load_golden("g2_forward.npz")
    net = TorchNet(weights, smpl["J"])
    pose, joints, vel, contact, r6d = net.forward(g["imu"], g[f"{tag}_lengths"].tolist())
    assert np.abs(joints - g[f"{tag}_joints"]).max() < TIGHT
    assert np.abs(r6d - g[f"{tag}_r6d"]).max() < TIGHT
    assert np.abs(vel - g[f"{tag}_vel"]).max() < TIGHT
    assert np.abs(contact - g[f"{tag}_contact"]).max() < TIGHT
    assert geodesic(pose, g[f"{tag}_pose"]).max() < TOL


def test_g9_full_motion_evaluator(smpl):
    """The evaluator restatement against the reference's own FullMotionEvaluator table (cv2 stand-in in the
    golden script: only |rotvec| is used)."""
    g = load_golden("g9_evaluator.npz")
    pp, pt = g["pose_p"].copy(), g["pose_t"].copy()
    pp[:, O.IGNORED] = np.eye(3)
    pt[:, O.IGNORED] = np.eye(3)
    e = O.full_motion_evaluator(pp, pt, smpl, g["tran_p"], g["tran_t"])
    assert np.abs(e - g["errs"]).max() / np.abs(g["errs"]).max() < 1e-6
    np.testing.assert_allclose(e, g["errs"], rtol=2e-5)


def test_g13_evaluate_pose_offline_and_online_tables(weights, smpl):
    """evaluate.py:57-65,96-103 with ONLINE=1, restated with the oracle's network and evaluator, against the tables the
    reference's own evaluate_pose printed for the same two sequences (golden G13): one model across both sequences (the
    velocity LSTM state and the last foot positions survive reset()), the online feed padded with 5 copies of the last
    frame and its first 5 outputs dropped."""
    g = load_golden("g13_evaluate_online.npz")
    net = O.OracleNet(weights, smpl["J"])
    sel = lambda e: np.stack([e[9], e[3], e[9], e[0] * 100, e[7] * 100, e[1] * 100, e[4] / 100, e[6]])   # evaluate.py:29
    off, on = [], []
    for k in range(int(g["n_seq"])):
        x, pose_t, tran_t = g[f"s{k}_imu"], g[f"s{k}_pose_t"].copy(), g[f"s{k}_tran_t"]
        pose_t[:, O.IGNORED] = np.eye(3)
        net.reset()
        pose_p, _, tran_p, _ = net.forward_offline(x[None], [x.shape[0]])
        pose_p = pose_p.reshape(-1, 24, 3, 3).copy()
        pose_p[:, O.IGNORED] = np.eye(3)
        off.append(sel(O.full_motion_evaluator(pose_p, pose_t, smpl, tran_p, tran_t)))
        frames = [net.forward_online(f) for f in np.concatenate((x, np.repeat(x[-1:], 5, axis=0)))]
        pose_o = np.stack([f[0] for f in frames])[5:].reshape(-1, 24, 3, 3).copy()
        tran_o = np.stack([f[2] for f in frames])[5:]
        pose_o[:, O.IGNORED] = np.eye(3)
        on.append(sel(O.full_motion_evaluator(pose_o, pose_t, smpl, tran_o, tran_t)))
    np.testing.assert_allclose(np.mean(off, axis=0), g["offline"], rtol=3e-4, atol=1e-4)
    np.testing.assert_allclose(np.mean(on, axis=0), g["online"], rtol=3e-4, atol=1e-4)


# ---- round 4: trained-like weights (G14) and the real mesh size (G15) ------------------------------------------------
@pytest.fixture(scope="module")
def weights_trained():
    from mobileposer_amd.synthetic import make_weights
    return make_weights(0, profile="trained")


def test_g14_trained_forward_ragged_mixed_combos(weights_trained, smpl):
    """forward on a ragged [6,60] batch, six different sensor combos, weights in the trained regime (saturated gates,
    recurrent gain > 1, forget bias + 1): the oracle stays within 1e-4 of the reference (measured 5e-6)."""
    g = load_golden("g14_trained.npz")
    net = O.OracleNet(weights_trained, smpl["J"])
    pose, joints, vel, contact = net.forward(g["imu"], g["lengths"].tolist())
    assert np.abs(net._last_r6d - g["r6d"]).max() < TOL
    assert np.abs(joints - g["joints"]).max() < TOL
    assert np.abs(vel - g["vel"]).max() < TOL
    assert np.abs(contact - g["contact"]).max() < TOL
    assert geodesic(pose, g["pose"]).max() < TOL
    h, c = net.velocity_rnn_state
    assert np.abs(h - g["vel_h"]).max() < TOL and np.abs(c - g["vel_c"]).max() < TOL
    # the regime is what it claims to be: cell states far outside (-1, 1)
    assert np.abs(g["vel_c"]).max() > 4.0


def test_g14_trained_offline_600_frames_pins_the_long_branch(weights_trained, smpl):
    """forward_offline at T = 600 > 512: the oracle's float64 running sum against net.py:154's fp32 re-summation."""
    g = load_golden("g14_trained.npz")
    net = O.OracleNet(weights_trained, smpl["J"])
    net.reset()
    pose, joints, tran, contact = net.forward_offline(g["off_imu"], [600])
    assert np.abs(joints - g["off_joints"]).max() < TOL
    assert np.abs(contact - g["off_contact"]).max() < TOL
    assert geodesic(pose, g["off_pose"]).max() < TOL
    assert np.abs(tran - g["off_tran"]).max() < 5e-5          # 1 mm is the bound; 4e-6 measured
    # and the short branch's arithmetic on the same data agrees with it (the two forms are interchangeable at 1e-5)
    j = joints[0].reshape(-1, 24, 3)
    short = O.translate_offline(j[:512], g_vel_of(net, g)[:512], contact[:512], net.floor_y)
    assert np.abs(short - g["off_tran"][:512]).max() < 5e-5


def g_vel_of(net, g):
    n2 = O.OracleNet(net.sd, net.J)
    _, _, vel, _ = n2.forward(g["off_imu"], [600])
    return vel[0]


def test_g14_trained_online_50_frames(weights_trained, smpl):
    g = load_golden("g14_trained.npz")
    net = O.OracleNet(weights_trained, smpl["J"])
    net.reset()
    for k, f in enumerate(g["on_imu"]):
        pose, joints, tran, contact = net.forward_online(f)
        assert geodesic(pose.reshape(24, 3, 3), g["on_pose"][k].reshape(24, 3, 3)).max() < TOL
        assert np.abs(joints[40] - g["on_joints40"][k]).max() < TOL
        assert np.abs(contact - g["on_contact"][k]).max() < TOL
        assert np.abs(tran - g["on_tran"][k]).max() < 1e-3, k
    h, c = net.velocity_rnn_state
    assert np.abs(h - g["on_vel_h"]).max() < TOL and np.abs(c - g["on_vel_c"]).max() < TOL


@pytest.mark.parametrize("tag", ["tr", "s1"])
def test_g14_all_twelve_combos(weights_trained, smpl, tag):
    """Row k of the batch keeps the devices of combo k (config.py:60-73, data.py:69-76); trained profile and a second seed."""
    from mobileposer_amd.synthetic import make_weights
    g = load_golden("g14_trained.npz")
    sd = weights_trained if tag == "tr" else make_weights(1)
    net = O.OracleNet(sd, smpl["J"])
    pose, joints, vel, contact = net.forward(g["c12_imu"], [40] * 12)
    assert np.abs(net._last_r6d - g[f"c12_{tag}_r6d"]).max() < TOL
    assert np.abs(joints - g[f"c12_{tag}_joints"]).max() < TOL
    assert np.abs(vel - g[f"c12_{tag}_vel"]).max() < TOL
    assert np.abs(contact - g[f"c12_{tag}_contact"]).max() < TOL


def test_g15_mesh_at_6890_vertices():
    """forward_kinematics(calc_mesh=True) without / with shape, the zero-pose body of a shape and pose blend shapes
    (articulate/model.py:77-92,208-240) on a 6890-vertex body: 26 full 256-vertex chunks and a 234-vertex tail."""
    from mobileposer_amd.synthetic import synthetic_smpl
    big = synthetic_smpl(n_vertex=6890)
    g = load_golden("g15_mesh6890.npz")
    _, jg, vg = O.forward_kinematics_mesh(g["pose"], big, tran=g["tran"])
    assert vg.shape == (3, 6890, 3)
    assert np.abs(jg - g["joint"]).max() < TIGHT and np.abs(vg - g["vert"]).max() < TIGHT
    _, jg, vg = O.forward_kinematics_shape(g["pose"], big, g["shape"], tran=g["tran"])
    assert np.abs(jg - g["shape_joint"]).max() < TIGHT and np.abs(vg - g["shape_vert"]).max() < TIGHT
    j0, v0 = O.shaped_body(big, g["shape"][:2])
    assert np.abs(j0 - g["zero_joint"]).max() < TIGHT and np.abs(v0 - g["zero_vert"]).max() < TIGHT
    _, jg, vg = O.forward_kinematics_shape(g["pose"], big, g["shape"][:1], tran=g["tran"], pose_blendshape=True)
    assert np.abs(jg - g["blend_joint"]).max() < TIGHT and np.abs(vg - g["blend_vert"]).max() < TIGHT
    _, _, vg = O.forward_kinematics_shape(g["pose"], big, None, pose_blendshape=True)
    assert np.abs(vg - g["blend_vert_noshape"]).max() < TIGHT
    # pose blend shapes do move vertices (the golden is not vacuous)
    _, _, plain = O.forward_kinematics_mesh(g["pose"], big)
    assert np.abs(plain - g["blend_vert_noshape"]).max() > 1e-3


def _same_nan_pattern_and_values(a, b, tol):
    assert a.shape == b.shape and (np.isnan(a) == np.isnan(b)).all()
    m = ~np.isnan(b)
    assert np.abs(a[m] - b[m]).max() < tol


def test_g16_nan_sample_follows_the_reference(smpl):
    """G16a: one NaN in one IMU sample.  The reference returns NaN joints / velocity / contact for every frame of that sequence
    and only there, and a finite pose everywhere (F.relu keeps the NaN, the bidirectional joints layers carry it both ways,
    angular.py:181 turns NaN rotations into 0): the oracle reproduces the pattern and the finite values."""
    from mobileposer_amd.synthetic import make_weights
    g = load_golden("g16_corners.npz")
    net = O.OracleNet(make_weights(0), smpl["J"])
    with np.errstate(all="ignore"):
        pose, joints, vel, contact = net.forward(g["nan_imu"], [24] * 5)
    assert np.isnan(g["nan_joints"][2]).all() and not np.isnan(g["nan_joints"][[0, 1, 3, 4]]).any()
    assert not np.isnan(g["nan_pose"]).any()
    _same_nan_pattern_and_values(joints, g["nan_joints"], TOL)
    _same_nan_pattern_and_values(vel.reshape(g["nan_vel"].shape), g["nan_vel"], TOL)
    _same_nan_pattern_and_values(contact, g["nan_contact"], TOL)
    # (the poisoned sequence's "rotations" are all-zero matrices: compared as numbers, an angle between them means nothing)
    assert not np.isnan(pose).any() and np.abs(pose - g["nan_pose"]).max() < 1e-5


def test_g16_one_frame_calls_on_a_carried_state(weights_trained, smpl):
    """G16b: five forward() calls of ONE frame each, velocity state carried (velocity.py:45-48), trained-regime weights."""
    g = load_golden("g16_corners.npz")
    net = O.OracleNet(weights_trained, smpl["J"])
    for k in range(5):
        pose, joints, vel, contact = net.forward(g["one_imu"][:, k:k + 1], [1] * 4)
        assert np.abs(joints - g[f"one_joints{k}"]).max() < TOL
        assert np.abs(vel.reshape(g[f"one_vel{k}"].shape) - g[f"one_vel{k}"]).max() < TOL
        assert np.abs(contact - g[f"one_contact{k}"]).max() < TOL
    h, c = net.velocity_rnn_state
    assert np.abs(h - g["one_vel_h"]).max() < TOL and np.abs(c - g["one_vel_c"]).max() < TOL


def test_g17_single_sequence_trained_regime_band(weights_trained, smpl):
    """Golden G17 (round 6): forward_offline of ONE 2000-frame sequence on trained-regime weights, recorded from the reference
    (evaluate.py:54-58).  At this length the net is chaotic at fp32 resolution, so the file carries -- beside the reference's
    outputs -- the distance of each member of an ensemble of fp32 evaluations (the reference, this oracle, the oracle with fourteen
    permuted summation orders: oracle/ensemble.py -- 16 members) from the float64 result.  Here: the input regenerates from its seed, the
    oracle on this host lies inside that band, and it is as close to the reference's recorded outputs as the two distances from
    the float64 result allow.  (The GPU test of the same golden: tests/test_gpu_round6.py.)"""
    from mobileposer_amd import synthetic
    from oracle import ensemble as ENS
    g = load_golden("g17_single_sequence.npz")
    assert [str(m) for m in g["members"]] == ["reference", "oracle"] + ["perm%d" % k for k in range(14)] and tuple(str(o) for o in g["outputs"]) == ENS.OUTPUTS
    T, seed, combo = 2000, int(g["seeds"][0]), str(g["combos"][0])
    imu = synthetic.make_imu(1, T, seed=seed, combo=combo)
    tag = "T%d_s%d" % (T, seed)
    assert abs(float(imu.astype(np.float64).sum()) - float(g[tag + "_imu_sum"])) < 1e-9
    truth = ENS.offline_outputs(weights_trained, smpl["J"], imu, T, dtype=np.float64)
    got = ENS.offline_outputs(weights_trained, smpl["J"], imu, T)
    d = ENS.distance(got, truth)
    dist = g[tag + "_dist"]
    tol = {"r6d": 1e-4, "joints": 1e-4, "vel": 1e-4, "contact": 1e-4, "tran": 1e-3}
    sub = slice((T - 1) % int(g["stride"]), None, int(g["stride"]))
    for i, k in enumerate(ENS.OUTPUTS):
        assert d[k][0] <= max(tol[k], 2.0 * dist[:, i, 0].max()), (k, d[k], dist[:, i, 0])
        assert d[k][1] <= max(0.01 * tol[k], 2.0 * dist[:, i, 1].max()), (k, d[k], dist[:, i, 1])
        mine = got[k] if k in ("contact", "tran") else got[k][sub]
        assert np.abs(mine - g[tag + "_" + k]).max() <= 1.001 * (d[k][0] + dist[0, i, 0]) + 1e-7, k
    # every recorded member of every case is a finite, small distance: the band itself is sane (the widest case is 4e-3 on r6d)
    for TT in g["lengths"].tolist():
        for s in g["seeds"].tolist():
            dd = g["T%d_s%d_dist" % (TT, s)]
            assert dd.shape == (16, 5, 2) and np.isfinite(dd).all() and (dd[:, :4, 0] < 2e-2).all() and (dd[:, 4, 0] < 5e-3).all(), (TT, s)


def test_g18_lengths_none_is_time_major(weights, smpl):
    """Golden G18 (round 6): input_lengths=None.  nn.LSTM is built without batch_first (rnn.py:15) and only the packed path is
    batch-first (rnn.py:25), so the reference treats dim 0 of [B,T,60] as time (outputs differ from the batch-first
    reading by 3.8e-2).  forward twice (the carried velocity state has batch T) and forward_offline of [1,40,60]."""
    g = load_golden("g18_lengths_none.npz")
    ref = O.OracleNet(weights, smpl["J"])
    for call in (0, 1):
        pose, joints, vel, contact = ref.forward(g["imu"], None)
        assert pose.shape == g[f"c{call}_pose"].shape and joints.shape == g[f"c{call}_joints"].shape
        assert np.abs(joints - g[f"c{call}_joints"]).max() < 2e-5
        assert np.abs(np.asarray(vel).reshape(g[f"c{call}_vel"].shape) - g[f"c{call}_vel"]).max() < 2e-5
        assert np.abs(contact - g[f"c{call}_contact"]).max() < 2e-5
        assert geodesic(pose, g[f"c{call}_pose"]).max() < 2e-5
    h, c = ref.velocity_rnn_state
    assert h.shape == g["vel_h"].shape and np.abs(h - g["vel_h"]).max() < 2e-5 and np.abs(c - g["vel_c"]).max() < 2e-5
    # ... and it is NOT what the batch-first reading gives
    other = O.OracleNet(weights, smpl["J"]).forward(g["imu"], [25, 25, 25])
    assert np.abs(other[1] - g["c0_joints"]).max() > 1e-3


def test_g19_submodules_as_the_reference_calls_them(weights):
    """Golden G19 (round 6): model.joints / model.pose / model.foot_contact / model.velocity called directly (net.py:103-117), with
    lengths and with input_lengths=None (time-major, rnn.py:15,25), and velocity.forward_online twice on its carried state
    (velocity.py:45-48)."""
    g = load_golden("g19_submodules.npz")
    lengths = g["lengths"].tolist()
    for name in ("joints", "pose", "foot_contact", "velocity"):
        y, _ = O.rnn_forward(weights, O.PREFIX[name], g[f"{name}_x"], lengths)
        assert y.shape == g[f"{name}_y"].shape and np.abs(y - g[f"{name}_y"]).max() < TIGHT, name
        yn, _ = O.rnn_forward(weights, O.PREFIX[name], g[f"{name}_x"], None)
        assert yn.shape == g[f"{name}_y_none"].shape and np.abs(yn - g[f"{name}_y_none"]).max() < TIGHT, name
        assert np.abs(yn - g[f"{name}_y"]).max() > 1e-3, name          # ... which is NOT the batch-first reading
    for tag, lens in (("online", lengths), ("online_none", None)):
        state = None
        for call in (0, 1):
            y, state = O.rnn_forward(weights, O.PREFIX["velocity"], g["velocity_x"], lens, state)
            assert np.abs(y - g[f"{tag}{call}"]).max() < TIGHT, (tag, call)
        assert state[0].shape == g[f"{tag}_h"].shape
        assert np.abs(state[0] - g[f"{tag}_h"]).max() < TIGHT and np.abs(state[1] - g[f"{tag}_c"]).max() < TIGHT


def test_g20_rotation_kinematics(smpl):
    """Golden G20 (round 6): ParametricModel.forward_kinematics_R / inverse_kinematics_R on their own (articulate/model.py:126-164;
    the latter is what MobilePoserNet.global_to_local_pose is bound to, net.py:38) on random rotations, and the round trip."""
    g = load_golden("g20_rotation_kinematics.npz")
    Rg, _ = O.forward_kinematics(g["R"], smpl["J"])
    assert np.abs(Rg - g["fk_R"]).max() < TIGHT
    assert np.abs(O.inverse_kinematics_R(g["R"]) - g["ik_R"]).max() < TIGHT
    assert np.abs(O.inverse_kinematics_R(g["fk_R"]) - g["ik_of_fk"]).max() < TIGHT
    assert np.abs(g["ik_of_fk"] - g["R"]).max() < 1e-5                      # IK(FK(R)) = R

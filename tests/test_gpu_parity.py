"""GPU parity tests: the HIP path, called through the C ABI (ctypes facade), against
  (1) golden vectors recorded from the reference itself (tests/golden/*.npz),
  (2) the CPU oracle on seeded inputs, at sizes the oracle finishes in seconds,
  (3) size-independent properties at BASELINE.json's full size (256 x 125).
Tolerances (BASELINE.json north_star): 1e-4 on joint angles and raw network outputs, 1 mm on root translation.
"""
import numpy as np
import pytest

from conftest import cu, geodesic, load_golden, npy

pytestmark = pytest.mark.gpu

TOL = 1e-4
TOL_TRAN = 1e-3


def test_native_library_is_what_runs(net):
    import ctypes
    assert net._h is not None and isinstance(net._lib, ctypes.CDLL)
    assert abs(net.floor_y - float(net.j[10:12, 1].min())) < 1e-7


@pytest.mark.parametrize("name", ["joints", "pose", "foot_contact", "velocity"])
def test_g1_rnn_ragged_golden(torch_mod, net, name):
    g = load_golden("g1_rnn.npz")
    lengths = g["lengths"].tolist()
    y, (h, c) = net.rnn_forward(name, cu(torch_mod, g[f"{name}_x"]), lengths)
    assert np.abs(npy(y) - g[f"{name}_y"]).max() < TOL
    assert np.abs(npy(h) - g[f"{name}_h"]).max() < TOL
    assert np.abs(npy(c) - g[f"{name}_c"]).max() < TOL
    y2, (h2, c2) = net.rnn_forward(name, cu(torch_mod, g[f"{name}_x"]), lengths, (h, c))
    assert np.abs(npy(y2) - g[f"{name}_y2"]).max() < TOL
    assert np.abs(npy(h2) - g[f"{name}_h2"]).max() < TOL
    assert np.abs(npy(c2) - g[f"{name}_c2"]).max() < TOL


@pytest.mark.parametrize("tag", ["eq", "rag"])
def test_g2_forward_golden(torch_mod, net, tag):
    g = load_golden("g2_forward.npz")
    pose, joints, vel, contact, r6d = net.forward(cu(torch_mod, g["imu"]), g[f"{tag}_lengths"].tolist(), return_r6d=True)
    assert tuple(pose.shape) == g[f"{tag}_pose"].shape
    assert np.abs(npy(joints) - g[f"{tag}_joints"]).max() < TOL
    assert np.abs(npy(vel) - g[f"{tag}_vel"]).max() < TOL
    assert np.abs(npy(contact) - g[f"{tag}_contact"]).max() < TOL
    assert np.abs(npy(r6d) - g[f"{tag}_r6d"]).max() < TOL
    assert geodesic(npy(pose), g[f"{tag}_pose"]).max() < TOL
    h, c = net.velocity.rnn_state
    assert np.abs(npy(h) - g[f"{tag}_vel_h"]).max() < TOL and np.abs(npy(c) - g[f"{tag}_vel_c"]).max() < TOL


def test_g3_r6d_ik_golden_with_degenerate_rows(torch_mod, net):
    g = load_golden("g3_r6d_ik.npz")
    pose = npy(net._reduced_global_to_full(cu(torch_mod, g["r6d"])))
    assert not np.isnan(pose).any()
    assert np.abs(pose - g["pose"]).max() < 1e-5


def test_g4_offline_golden_and_stale_velocity_state(torch_mod, net):
    g = load_golden("g4_offline.npz")
    for tag, x in (("a", g["imu_a"]), ("b", g["imu_b"]), ("a_again", g["imu_a"])):
        net.reset()
        pose, joints, tran, contact = net.forward_offline(cu(torch_mod, x), [x.shape[1]])
        assert tuple(pose.shape) == g[f"{tag}_pose"].shape and tuple(tran.shape) == g[f"{tag}_tran"].shape
        assert geodesic(npy(pose), g[f"{tag}_pose"]).max() < TOL, tag
        assert np.abs(npy(joints) - g[f"{tag}_joints"]).max() < TOL
        assert np.abs(npy(contact) - g[f"{tag}_contact"]).max() < TOL
        assert np.abs(npy(tran) - g[f"{tag}_tran"]).max() < TOL_TRAN, tag
    # Q1: clearing the velocity state restores the first answer
    net.reset()
    net.velocity.rnn_state = None
    _, _, tran, _ = net.forward_offline(cu(torch_mod, g["imu_a"]), [g["imu_a"].shape[1]])
    assert np.abs(npy(tran) - g["a_cleared_tran"]).max() < TOL_TRAN
    assert np.abs(npy(tran) - g["a_again_tran"]).max() > 1e-5


def test_g5_online_golden(torch_mod, net):
    g = load_golden("g5_online.npz")
    net.reset()
    for k, f in enumerate(g["imu"]):
        pose, joints, tran, contact = net.forward_online(cu(torch_mod, f))
        assert tuple(pose.shape) == (24, 9) and tuple(joints.shape) == (45, 72)
        assert geodesic(npy(pose).reshape(24, 3, 3), g["pose"][k].reshape(24, 3, 3)).max() < TOL, k
        assert np.abs(npy(joints)[40] - g["joints40"][k]).max() < TOL
        assert np.abs(npy(contact) - g["contact"][k]).max() < TOL
        assert np.abs(npy(tran) - g["tran"][k]).max() < TOL_TRAN, k
    h, c = net.velocity.rnn_state
    assert np.abs(npy(h) - g["vel_h"]).max() < TOL and np.abs(npy(c) - g["vel_c"]).max() < TOL


def test_g6_fk_golden(torch_mod, net):
    g = load_golden("g6_fk.npz")
    Rg, jg = net.forward_kinematics(cu(torch_mod, g["pose"]))
    assert np.abs(npy(Rg) - g["R_global"]).max() < 1e-5
    assert np.abs(npy(jg) - g["joint"]).max() < 1e-5
    _, jg2 = net.bodymodel.forward_kinematics(cu(torch_mod, g["pose"]), tran=cu(torch_mod, g["tran"]))
    assert np.abs(npy(jg2) - g["joint_tran"]).max() < 1e-5


def test_translate_offline_vs_oracle_ragged(torch_mod, net):
    """K6 alone on crafted inputs: both feet take turns, the floor clamp fires, ragged lengths."""
    from oracle import mp_oracle as O
    rng = np.random.Generator(np.random.PCG64(77))
    B, T = 5, 300
    joints = (rng.standard_normal((B, T, 72)) * 0.05).astype(np.float32)
    joints[:, :, 31] -= 0.9
    joints[:, :, 34] -= 0.9
    vel = (rng.standard_normal((B, T, 72)) * 0.3).astype(np.float32)
    contact = (rng.standard_normal((B, T, 2)) * 2.0).astype(np.float32)
    lengths = [300, 1, 17, 300, 123]
    tran = torch_mod.empty(B, T, 3, device="cuda")
    import ctypes as C
    net.translate_offline_into(cu(torch_mod, joints), cu(torch_mod, vel), cu(torch_mod, contact), (C.c_int32 * B)(*lengths), tran)
    tran = npy(tran)
    clamped = 0
    for b in range(B):
        L = lengths[b]
        ref = O.translate_offline(joints[b, :L].reshape(L, 24, 3), vel[b, :L], contact[b, :L], net.floor_y)
        assert np.abs(tran[b, :L] - ref).max() < 1e-4, b
        if L < T:
            assert np.abs(tran[b, L:] - ref[-1]).max() < 1e-4
        foot = ref[:, 1] + joints[b, :L].reshape(L, 24, 3)[:, 10:12, 1].min(axis=1)
        clamped += int((np.abs(foot - net.floor_y) < 1e-5).sum())
    assert clamped > 20


def test_forward_vs_oracle_medium(torch_mod, net, weights, smpl):
    """Seeded 24 x 60 batch with ragged lengths against the oracle (all four outputs + translation)."""
    from mobileposer_amd import synthetic
    from oracle import mp_oracle as O
    B, T = 24, 60
    imu = synthetic.make_imu(B, T, seed=3)
    lengths = [T] * B
    for b, L in ((1, 7), (5, 59), (17, 1), (23, 33)):
        lengths[b] = L
    pose, joints, vel, contact, r6d = net.forward(cu(torch_mod, imu), lengths, return_r6d=True)
    ref = O.OracleNet(weights, smpl["J"])
    rpose, rjoints, rvel, rcontact = ref.forward(imu, lengths)
    assert np.abs(npy(joints) - rjoints).max() < TOL
    assert np.abs(npy(vel) - rvel).max() < TOL
    assert np.abs(npy(contact) - rcontact).max() < TOL
    assert np.abs(npy(r6d) - ref._last_r6d).max() < TOL
    assert geodesic(npy(pose), rpose).max() < TOL


@pytest.mark.parametrize("B,T", [(72, 40), (128, 33)])
def test_half_chip_batches_vs_oracle(torch_mod, net, weights, smpl, B, T):
    """64 < B <= 128 (on the exact-fp32 path: pose layer 0 on 16 slices, then pose layer 1 on 8 slices beside the velocity
    layers that carry the foot-contact layers -- DESIGN.md section 4 and profiles/NOTES_r01-r03.md section 4), ragged, two calls so that the second one
    starts from a carried velocity state, against the oracle: network outputs, poses and translation of every sequence."""
    from mobileposer_amd import synthetic
    from oracle import mp_oracle as O
    imu = synthetic.make_imu(B, T, seed=B + T)
    lengths = [T - (7 * b) % (T - 1) for b in range(B)]
    lengths[0] = T
    ref = O.OracleNet(weights, smpl["J"])
    for call in range(2):
        pose, joints, vel, contact, r6d = net.forward(cu(torch_mod, imu), lengths, return_r6d=True)
        rpose, rjoints, rvel, rcontact = ref.forward(imu, lengths)
        for b in range(B):                                     # (rows past a sequence's length are padding on both sides)
            n = lengths[b]
            assert np.abs(npy(joints)[b, :n] - rjoints[b, :n]).max() < TOL, (call, b)
            assert np.abs(npy(vel)[b, :n] - rvel[b, :n]).max() < TOL, (call, b)
            assert np.abs(npy(contact)[b, :n] - rcontact[b, :n]).max() < TOL, (call, b)
            assert np.abs(npy(r6d)[b, :n] - ref._last_r6d[b, :n]).max() < TOL, (call, b)
        assert geodesic(npy(pose), rpose).max() < TOL, call
    assert net.device_error() == 0 and net.recovery_count == 0


def test_full_size_vs_oracle_and_properties(torch_mod, net, weights, smpl):
    """BASELINE config: 256 x 125.  Oracle comparison on the whole batch (network outputs, all 256 translation rows, FK of
    all 32 000 frames) plus size-independent properties: batch-permutation equivariance (sequences are independent),
    determinism."""
    import ctypes as C
    from mobileposer_amd import synthetic
    from oracle import mp_oracle as O
    B, T = 256, 125
    imu = synthetic.make_imu(B, T, seed=1)
    x = cu(torch_mod, imu)
    lengths = [T] * B
    net.reset_all()
    pose, joints, vel, contact = net.forward(x, lengths)
    ref = O.OracleNet(weights, smpl["J"])
    rpose, rjoints, rvel, rcontact = ref.forward(imu, lengths)
    assert np.abs(npy(joints) - rjoints).max() < TOL
    assert np.abs(npy(vel) - rvel).max() < TOL
    assert np.abs(npy(contact) - rcontact).max() < TOL
    assert geodesic(npy(pose), rpose).max() < TOL
    # translation at full size through the batched solver
    tran = torch_mod.empty(B, T, 3, device="cuda")
    net.translate_offline_into(joints, vel.reshape(B, T, 72), contact, (C.c_int32 * B)(*lengths), tran)
    tran_h = npy(tran)
    for b in range(B):                                # every row of the batch
        rt = O.translate_offline(rjoints[b].reshape(T, 24, 3), rvel[b], rcontact[b], ref.floor_y)
        assert np.abs(tran_h[b] - rt).max() < TOL_TRAN, b
    # SMPL FK of the predicted pose at N = 32 000 (BASELINE configs[2]) against the oracle
    Rg, jg = net.forward_kinematics(pose)
    rRg, rjg = O.forward_kinematics(rpose, smpl["J"])
    assert geodesic(npy(Rg), rRg).max() < TOL
    assert np.abs(npy(jg) - rjg).max() < 1e-4
    # determinism: same call again (state cleared) is bitwise identical
    net.reset_all()
    pose2, joints2, vel2, contact2 = net.forward(x, lengths)
    assert torch_mod.equal(pose, pose2) and torch_mod.equal(joints, joints2) and torch_mod.equal(vel, vel2)
    # permutation equivariance: sequences never interact
    perm = torch_mod.randperm(B, generator=torch_mod.Generator().manual_seed(0)).cuda()
    net.reset_all()
    pose4, joints4, vel4, contact4 = net.forward(x[perm], lengths)
    assert torch_mod.equal(joints4, joints[perm]) and torch_mod.equal(vel4, vel[perm])
    assert torch_mod.equal(pose4.reshape(B, T, 24, 9), pose.reshape(B, T, 24, 9)[perm])


def test_ragged_equals_truncated(torch_mod, net):
    """Packed-sequence semantics (Q4): a short sequence inside a padded batch == the same sequence alone."""
    from mobileposer_amd import synthetic
    B, T, L = 4, 50, 23
    imu = synthetic.make_imu(B, T, seed=9)
    net.reset_all()
    _, joints, vel, contact, r6d = net.forward(cu(torch_mod, imu), [T, L, T, T], return_r6d=True)
    net.reset_all()
    _, joints1, vel1, contact1, r6d1 = net.forward(cu(torch_mod, imu[1:2, :L]), [L], return_r6d=True)
    assert np.abs(npy(joints[1, :L]) - npy(joints1[0])).max() < 1e-6
    assert np.abs(npy(r6d[1, :L]) - npy(r6d1[0])).max() < 1e-6
    assert np.abs(npy(vel[1, :L]) - npy(vel1)).max() < 1e-6
    # padded frames carry linear2.bias exactly
    b2 = net.state_dict()["joints.joints.linear2.bias"].numpy()
    assert np.abs(npy(joints[1, L:]) - b2).max() < 1e-7


def test_error_behaviour(torch_mod, net):
    from mobileposer_amd import synthetic
    x = cu(torch_mod, synthetic.make_imu(2, 10, seed=2))
    with pytest.raises(ValueError):
        net.rnn_forward("joints", x, None)         # Q3: a module's own entry needs lengths (forward / forward_offline
                                                   # reproduce the reference's time-major reading since round 6: golden G18)
    with pytest.raises(RuntimeError):
        net.forward(x, [10, 11])                   # length > T
    with pytest.raises(RuntimeError):
        net.forward(x, [9, 9])                     # max(lengths) != T: the reference's cat fails
    net.forward(x, [10, 10])
    x3 = cu(torch_mod, synthetic.make_imu(3, 10, seed=2))
    with pytest.raises(RuntimeError):
        net.forward(x3, [10, 10, 10])              # Q2: carried velocity state has another batch size
    net.velocity.rnn_state = None
    net.forward(x3, [10, 10, 10])


def test_multi_stream_equals_single_streams(torch_mod, weights, smpl):
    """K7: S concurrent streams ticked together == each stream run alone through forward_online."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    S, n = 3, 8
    frames = synthetic.make_imu(S, n, seed=41)
    with MobilePoserNet.from_numpy(weights, smpl, device="cuda:0") as multi:
        multi.stream_create(S)
        outs = [multi.stream_step(cu(torch_mod, frames[:, k])) for k in range(n)]
        for s in range(S):                 # two handles alive and ticking: the first process that did this died in hipGraphLaunch
            with MobilePoserNet.from_numpy(weights, smpl, device="cuda:0") as single:
                single.reset()
                for k in range(n):
                    pose, joints, root, contact = single.forward_online(cu(torch_mod, frames[s, k]))
                    assert np.abs(npy(outs[k][0][s]) - npy(pose)).max() < 1e-5
                    assert np.abs(npy(outs[k][2][s]) - npy(root)).max() < 1e-5
                    assert np.abs(npy(outs[k][3][s]) - npy(contact)).max() < 1e-5
                assert single.device_error() == 0
        assert multi.device_error() == 0


def test_persistent_and_step_recurrence_agree(torch_mod, net):
    """The persistent kernel (granule exchange) and the per-step kernels compute the same recurrence;
    only the fp32 summation order over K differs.  Also checks that no bounded wait timed out."""
    from mobileposer_amd import synthetic
    B, T = 40, 50
    imu = cu(torch_mod, synthetic.make_imu(B, T, seed=12))
    lengths = [T] * B
    lengths[3], lengths[17], lengths[39] = 5, 49, 1
    outs = {}
    for mode in (1, 0, 2, 3):       # fused persistent | per-step kernels | + two-layer wavefront velocity | split-bf16
        net.set_lstm_mode(mode)
        net.reset_all()
        outs[mode] = [t.clone() for t in net.forward(imu, lengths)]
        # carried velocity state must survive the mode as well (second call starts from the first call's state)
        outs[mode] += [t.clone() for t in net.forward(imu, lengths)]
        assert net.device_error() == 0
    net.set_lstm_mode(net.lstm_mode)
    for other in (0, 2, 3):
        for a, b in zip(outs[1], outs[other]):
            assert float((a - b).abs().max()) < 2e-5, other


def test_no_device_error_after_full_size(torch_mod, net):
    from mobileposer_amd import synthetic
    x = cu(torch_mod, synthetic.make_imu(256, 125, seed=1))
    for _ in range(3):
        net.reset_all()
        net.forward(x, [125] * 256)
    assert net.device_error() == 0


def test_evaluate_harness_offline_and_online(torch_mod, net, monkeypatch):
    """evaluate.py flow on a synthetic dataset: runs end to end, FK-consistent metrics, ONLINE branch included."""
    from mobileposer_amd.data import PoseDataset
    from mobileposer_amd.evaluate import PoseEvaluator, evaluate_pose, synthetic_dataset
    from mobileposer_amd.config import amass
    data = synthetic_dataset(n_seq=1, frames=40, seed=3)
    ds = PoseDataset(fold='test', evaluate='dip', data=data, fk=net.forward_kinematics,
                     combos=dict(list(amass.combos.items())[:2]))
    assert len(ds) == 2 and tuple(ds[0][0].shape) == (40, 60) and tuple(ds[0][2].shape) == (40, 24, 3)
    monkeypatch.setenv("ONLINE", "1")
    out = evaluate_pose(net, ds, verbose=False)
    assert tuple(out["offline"].shape) == (8, 2) and tuple(out["online"].shape) == (8, 2)
    assert torch_mod.isfinite(out["offline"][[0, 1, 3, 6]]).all()
    # identical prediction and ground truth -> zero errors
    ev = PoseEvaluator(net)
    pose = data["pose"][0].cuda()
    e = ev.eval(pose, pose, tran_p=data["tran"][0], tran_t=data["tran"][0])
    assert float(e[[0, 1, 3, 4], 0].abs().max()) < 1e-3
    assert net.device_error() == 0


def test_large_ragged_batch_chunked_launches(torch_mod, net, weights, smpl):
    """B = 300 (> 256 and not a multiple of 16): the persistent layers run as several co-resident launches with a
    partial last slab; spot-check rows against the oracle."""
    from mobileposer_amd import synthetic
    from oracle import mp_oracle as O
    B, T = 300, 20
    imu = synthetic.make_imu(B, T, seed=17)
    lengths = [T] * B
    lengths[0], lengths[255], lengths[256], lengths[299] = 3, 11, 20, 7
    net.reset_all()
    pose, joints, vel, contact = net.forward(cu(torch_mod, imu), lengths)
    assert net.device_error() == 0
    rows = [0, 15, 16, 255, 256, 288, 299]
    for r in rows:                                   # one sequence at a time: lengths differ
        L = lengths[r]
        ro = O.OracleNet(weights, smpl["J"])
        _, rj, rv, rc = ro.forward(imu[r:r + 1, :L], [L])
        assert np.abs(npy(joints[r, :L]) - rj[0]).max() < TOL, r
        assert np.abs(npy(vel[r, :L]) - rv[0]).max() < TOL, r
        assert np.abs(npy(contact[r, :L]) - rc[0]).max() < TOL, r


def test_many_shapes_evict_plans(torch_mod, net):
    """evaluate.py-style stream of different sequence lengths: plans and graphs are evicted, results stay right."""
    from mobileposer_amd import synthetic
    imu = synthetic.make_imu(1, 48, seed=23)
    net.reset_all()
    first = net.forward(cu(torch_mod, imu[:, :12]), [12])[1].clone()
    for T in range(13, 43):                          # 30 more shapes > kMaxPlans (24 since round 5)
        net.reset_all()
        net.forward(cu(torch_mod, imu[:, :T]), [T])
    net.reset_all()
    again = net.forward(cu(torch_mod, imu[:, :12]), [12])[1]
    assert torch_mod.equal(first, again)
    assert net.device_error() == 0


def test_g6_fk_mesh_and_g9_evaluator_golden(torch_mod, net):
    """mp_fk_mesh (LBS) against the reference's vertices, and the evaluator error table against the reference's."""
    from mobileposer_amd.evaluate import FullMotionEvaluator
    g = load_golden("g6_fk.npz")
    Rg, jg, vg = net.forward_kinematics(cu(torch_mod, g["pose"]), cu(torch_mod, g["tran"]), calc_mesh=True)
    assert np.abs(npy(jg) - g["joint_tran"]).max() < 1e-5
    assert np.abs(npy(vg) - g["vert_tran"]).max() < 1e-5
    g9 = load_golden("g9_evaluator.npz")
    pp, pt = cu(torch_mod, g9["pose_p"]).clone(), cu(torch_mod, g9["pose_t"]).clone()
    ign = [0, 7, 8, 10, 11, 20, 21, 22, 23]
    pp[:, ign] = torch_mod.eye(3, device="cuda")
    pt[:, ign] = torch_mod.eye(3, device="cuda")
    ev = FullMotionEvaluator(net, joint_mask=[2, 5, 16, 20], fps=30)
    errs = npy(ev(pp, pt, tran_p=cu(torch_mod, g9["tran_p"]), tran_t=cu(torch_mod, g9["tran_t"])))
    np.testing.assert_allclose(errs, g9["errs"], rtol=2e-4, atol=1e-5)
    # the same through the `ignored` argument of the kernel (what PoseEvaluator.eval passes) on the unmasked poses
    ev2 = FullMotionEvaluator(net, joint_mask=[2, 5, 16, 20], fps=30, ignored=ign)
    errs2 = npy(ev2(cu(torch_mod, g9["pose_p"]), cu(torch_mod, g9["pose_t"]), tran_p=cu(torch_mod, g9["tran_p"]),
                    tran_t=cu(torch_mod, g9["tran_t"])))
    assert np.array_equal(errs, errs2)
    # no mask: {0, NaN} rows like torch.zeros(1); a sequence shorter than the 1-s window / the jerk stencil: NaN rows
    e3 = npy(FullMotionEvaluator(net, fps=30)(pp[:20], pt[:20]))
    assert (e3[7:, 0] == 0).all() and np.isnan(e3[7:, 1]).all() and np.isnan(e3[6]).all() and np.isfinite(e3[:6]).all()
    e4 = npy(FullMotionEvaluator(net, fps=30)(pp[:3], pt[:3]))
    assert np.isnan(e4[4:7]).all() and np.isfinite(e4[:4]).all()
    # r6d -> rotation matrix of the ground-truth side (evaluate.py:60) against the reference's function (golden G3 `rot`)
    g3 = load_golden("g3_r6d_ik.npz")
    rot = npy(net.r6d_to_rotation_matrix(cu(torch_mod, g3["r6d"])))
    assert np.abs(rot - g3["rot"].reshape(-1, 3, 3)).max() < 1e-6 and not np.isnan(rot).any()


def test_hidden_state_transports_agree(torch_mod, net):
    """The same-XCD (L2) transport and the any-placement (write-through) transport of the persistent kernels carry
    the same values: bitwise-identical outputs, for a full-chip batch and for a single small cluster."""
    from mobileposer_amd import synthetic
    for B, T in ((256, 30), (3, 40)):
        x = cu(torch_mod, synthetic.make_imu(B, T, seed=29))
        lengths = [T] * B
        outs = []
        for remote in (False, True):
            net.set_transport(remote)
            net.reset_all()
            outs.append([t.clone() for t in net.forward(x, lengths)])
            assert net.device_error() == 0
        net.set_transport(False)
        for a, b in zip(*outs):
            assert torch_mod.equal(a, b)


def test_split_bf16_kernel_variants_agree(torch_mod, weights, smpl, monkeypatch):
    """The two split-bf16 layer kernels (mp_lstm_x3: eight 256-register waves per workgroup; mp_lstm_x3w: four
    512-register waves, inline-asm MFMAs with AccVGPR operands) do the same arithmetic in the same order: whichever of them runs the K_in = 256 /
    K_in = 512 layers (MP_VARIANT x3w bit mask, default 2), the outputs are bitwise identical -- full-chip batch, ragged
    lengths, and a continued velocity state.  (This is also the check on the hand-placed MFMA wait states of mp_lstm_x3w.)"""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    B, T = 256, 40
    x = cu(torch_mod, synthetic.make_imu(B, T, seed=31))
    rng = np.random.default_rng(5)
    lengths = [int(v) for v in rng.integers(1, T + 1, size=B)]
    lengths[0] = T
    outs = {}
    for mask in (0, 1, 2, 3):
        monkeypatch.setenv("MP_VARIANT", "x3w=%d" % mask)
        n = MobilePoserNet.from_numpy(weights, smpl, device="cuda:0")
        n.set_lstm_mode(3)
        o1 = [t.clone() for t in n.forward(x, lengths)]
        o2 = [t.clone() for t in n.forward(x, lengths)]          # second call: velocity continues from its state (Q1)
        assert n.device_error() == 0
        outs[mask] = o1 + o2
        n.close()
    for mask in (1, 2, 3):
        for a, b in zip(outs[0], outs[mask]):
            assert torch_mod.equal(a, b), "x3w=%d differs from x3w=0 by %g" % (mask, float((a - b).abs().max()))


@pytest.mark.parametrize("B,T", [(1, 1), (1, 2), (17, 3), (2, 45)])
def test_tiny_shapes_vs_oracle(torch_mod, net, weights, smpl, B, T):
    """Edge shapes: single frame, single sequence, a slab with one valid row, the online window length."""
    from mobileposer_amd import synthetic
    from oracle import mp_oracle as O
    imu = synthetic.make_imu(B, T, seed=100 + B + T)
    lengths = [T] * B
    if B > 2:
        lengths[B // 2] = 1
    net.reset_all()
    pose, joints, tran, contact = net.forward_offline(cu(torch_mod, imu), lengths)
    assert net.device_error() == 0
    for b in range(B):
        L = lengths[b]
        ref = O.OracleNet(weights, smpl["J"])
        rp, rj, rt, rc = ref.forward_offline(imu[b:b + 1, :L], [L])
        jb = npy(joints[b, :L]) if B > 1 else npy(joints[0, :L])
        tb = npy(tran[b, :L]) if B > 1 else npy(tran[:L])
        cb = npy(contact[b, :L]) if B > 1 else npy(contact[:L])
        pb = npy(pose).reshape(B, T, 24, 3, 3)[b, :L]
        assert np.abs(jb - rj[0]).max() < TOL
        assert np.abs(cb - rc).max() < TOL
        assert np.abs(tb - rt).max() < TOL_TRAN
        assert geodesic(pb, rp).max() < TOL


def test_live_session_feeds_stream_step(torch_mod, weights, smpl):
    """Live front-end (calibration + frame formation) -> GPU streaming tick == forward_online on the same frames."""
    from mobileposer_amd import live
    from mobileposer_amd.net import MobilePoserNet
    rng = np.random.default_rng(4)
    S, n = 2, 4
    cals = [live.Calibration.from_measurements(torch_mod.from_numpy(rng.standard_normal(4)).float(),
                                               torch_mod.from_numpy(rng.standard_normal((5, 4))).float(),
                                               torch_mod.from_numpy(rng.standard_normal((5, 3))).float()) for _ in range(S)]
    quats = rng.standard_normal((n, S, 5, 4)).astype(np.float32)
    accs = (rng.standard_normal((n, S, 5, 3)) * 3).astype(np.float32)
    model = MobilePoserNet.from_numpy(weights, smpl)
    sess = live.LiveSession(model, cals)
    singles = [MobilePoserNet.from_numpy(weights, smpl) for _ in range(S)]
    for k in range(n):
        pose, root, packets = sess.tick(quats[k], accs[k])
        assert len(packets) == S and packets[0].endswith(b"$")
        for s in range(S):
            frame = live.form_frame(cals[s], torch_mod.from_numpy(quats[k, s])[None], torch_mod.from_numpy(accs[k, s])[None])[0]
            p1, _, r1, _ = singles[s].forward_online(frame.cuda())
            assert np.abs(npy(pose[s]) - npy(p1)).max() < 1e-5 and np.abs(npy(root[s]) - npy(r1)).max() < 1e-5
    for m in [model] + singles:
        m.close()


def test_soak_bitwise_stable_under_concurrency(torch_mod, net):
    """Glitch detector (profiles/NOTES_r01-r03.md 4.3): the whole captured forward + FK + solver -- LSTM layers, GEMMs, IK, FK and the
    solver running beside each other on four streams -- must give bit-identical outputs every time."""
    from mobileposer_amd import synthetic
    B, T = 256, 125
    x = cu(torch_mod, synthetic.make_imu(B, T, seed=41))
    lengths = [T] * B
    ref = None
    for it in range(12):
        net.reset_all()
        outs = [t.clone() for t in net.forward_offline(x, lengths)]
        if ref is None:
            ref = outs
        else:
            for a, b in zip(ref, outs):
                assert torch_mod.equal(a, b), it
    assert net.device_error() == 0


def test_small_batch_schedules_agree(torch_mod, weights, smpl, monkeypatch):
    """Batches whose pose + velocity + foot-contact launches fit the chip together run the three blocks side by side
    (MP_VARIANT wide, default on): bitwise the same outputs as the serial schedule.  Exact-fp32 layers of small batches use 16
    slices per slab instead of 8 (MP_VARIANT slices16, default on): another summation order, same values to fp32 noise."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    outs = {}
    for wide, s16 in ((1, 1), (0, 1), (1, 0)):
        monkeypatch.setenv("MP_VARIANT", "wide=%d,slices16=%d" % (wide, s16))
        with MobilePoserNet.from_numpy(weights, smpl) as n:
            o = []
            for mode in (1, 3):
                n.set_lstm_mode(mode)
                for B, T in ((1, 200), (40, 50), (64, 30)):
                    x = cu(torch_mod, synthetic.make_imu(B, T, seed=B))
                    L = [T] * B
                    if B > 1:
                        L[B // 2] = max(1, T // 3)
                    n.reset_all()
                    o += [t.clone() for t in n.forward_offline(x, L)]
                    o += [t.clone() for t in n.forward_offline(x, L)]      # carried velocity state
            assert n.device_error() == 0
        outs[(wide, s16)] = o
    for a, b in zip(outs[(1, 1)], outs[(0, 1)]):
        assert torch_mod.equal(a, b)
    for a, b in zip(outs[(1, 1)], outs[(1, 0)]):
        assert float((a - b).abs().max()) < 5e-6


def test_half_chip_schedules_agree(torch_mod, weights, smpl, monkeypatch):
    """64 < B <= 128, exact-fp32 operands.  Default (schedule 4, MP_VARIANT late_pair): pose layer 0 on 16 slices with the chip
    to itself, then pose layer 1 on 8 slices (half the chip) beside the velocity layers, the foot-contact layers riding in the
    velocity workgroups.  late_pair=0 (schedules 2 / 3, MP_VARIANT half): both pose layers on 8 slices beside velocity (and foot
    contact, B <= 96; after velocity otherwise).  Every cluster on an XCD chosen by the host.  half=0: the serial schedule.
    Same values to fp32 noise (8 against 16 slices, rider against stand-alone foot-contact kernel: other summation orders);
    where two schedules use the same kernels -- joints always; velocity / foot contact of schedules 2 / 3 and the serial one --
    the outputs must agree bitwise."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    outs = {}
    for variant in ("late_pair=1", "late_pair=0", "half=0"):
        monkeypatch.setenv("MP_VARIANT", variant)
        with MobilePoserNet.from_numpy(weights, smpl) as n:
            n.set_lstm_mode(1)                                  # the half-chip schedules exist for exact-fp32 operands only
            o = []
            for B, T in ((80, 24), (96, 20), (128, 16), (100, 30), (65, 40)):
                x = cu(torch_mod, synthetic.make_imu(B, T, seed=B))
                L = [T] * B
                L[B // 2] = max(1, T // 3)
                L[-1] = max(1, T - 2)
                n.reset_all()
                o += [t.clone() for t in n.forward_offline(x, L)]
                o += [t.clone() for t in n.forward_offline(x, L)]          # carried velocity state
            assert n.device_error() == 0 and n.recovery_count == 0
        outs[variant] = o
    for i, (a, b, c) in enumerate(zip(outs["late_pair=1"], outs["late_pair=0"], outs["half=0"])):
        if i % 4 == 0:                                         # pose: 8-slice against 16-slice kernels
            assert float((b - c).abs().max()) < 5e-6 and float((a - c).abs().max()) < 5e-6
        elif i % 4 == 1:                                       # joints: the same kernels in all three
            assert torch_mod.equal(b, c) and torch_mod.equal(a, c)
        else:                                                  # translation, contact: rider in schedule 4, the same kernels otherwise
            assert torch_mod.equal(b, c)
            assert float((a - c).abs().max()) < 5e-6
    assert any(not torch_mod.equal(a, b) for a, b in zip(outs["late_pair=1"], outs["late_pair=0"]))   # (schedule 4 did run)


def test_half_chip_streaming_ticks_vs_oracle(torch_mod, net, weights, smpl):
    """S = 100 concurrent streams (64 < S <= 128: on the exact-fp32 path every tick runs schedule 4 with the velocity state
    carried in place from tick to tick), 8 ticks, four of the streams followed by the oracle's forward_online."""
    from mobileposer_amd import synthetic
    from oracle import mp_oracle as O
    S, n = 100, 8
    watch = [0, 63, 64, 99]
    frames = synthetic.make_imu(S, n, seed=101)
    refs = {s: O.OracleNet(weights, smpl["J"]) for s in watch}
    net.reset_all()
    net.stream_create(S)
    for k in range(n):
        pose, joints, root, contact = net.stream_step(cu(torch_mod, frames[:, k]))
        for s in watch:
            rp, rj, rr, rc = refs[s].forward_online(frames[s, k])
            assert geodesic(npy(pose[s]).reshape(24, 3, 3), rp.reshape(24, 3, 3)).max() < TOL, (s, k)
            assert np.abs(npy(joints[s]) - rj).max() < TOL, (s, k)
            assert np.abs(npy(contact[s]) - rc).max() < TOL, (s, k)
            assert np.abs(npy(root[s]) - rr).max() < TOL_TRAN, (s, k)
    assert net.device_error() == 0 and net.recovery_count == 0


def test_g11_evaluate_pose_table_and_translation_statistics(torch_mod, net):
    """evaluate_pose (evaluate.py:39-107) against the reference's own run on canned predictions (golden G11): the 8 x 2
    table -- aggregated with mean(), so the sequence shorter than one second turns the 1-s distance row into NaN exactly
    as upstream -- and the evaluate_tran list.  FK / skinning / angles / reductions run on the GPU."""
    from mobileposer_amd.data import rotation_matrix_to_r6d
    from mobileposer_amd.evaluate import evaluate_pose
    g = load_golden("g11_evaluate.npz")
    n_seq = int(g["n_seq"])

    class Canned:                                   # the model under evaluation is not the point here: canned predictions
        device, n_vertex = net.device, net.n_vertex
        forward_kinematics = staticmethod(net.forward_kinematics)
        eval_metrics = staticmethod(net.eval_metrics)
        r6d_to_rotation_matrix = staticmethod(net.r6d_to_rotation_matrix)
        k = -1

        def eval(self):
            return self

        def reset(self):
            self.k += 1

        def forward_offline(self, x, lengths):
            return (cu(torch_mod, g[f"s{self.k}_pose_p"]), None, cu(torch_mod, g[f"s{self.k}_tran_p"]), None)

    ds = [(torch_mod.from_numpy(g[f"s{k}_imu"]), rotation_matrix_to_r6d(torch_mod.from_numpy(g[f"s{k}_pose_t"])).reshape(-1, 144),
           None, torch_mod.from_numpy(g[f"s{k}_tran_t"])) for k in range(n_seq)]
    out = evaluate_pose(Canned(), ds, evaluate_tran=True, verbose=False)
    table = npy(out["offline"])
    assert np.isnan(table[7]).all() and np.isnan(g["table"][7]).all()
    np.testing.assert_allclose(table[:7], g["table"][:7], rtol=3e-4, atol=1e-4)
    np.testing.assert_allclose(out["tran"], g["tran_errors"], rtol=1e-4, atol=1e-6)


def test_fused_fp32_kernels_match_the_per_step_kernels_on_ragged_launch_groups(torch_mod, weights, smpl):
    """The fused layer kernels (mode 1: the four-wave 8-slice kernels, the two-layer velocity wavefront and its riders) against
    the per-step kernels (mode 0: input-projection GEMM + one launch per time step, no cross-workgroup hand-off): full-chip
    batch, several launch groups with an odd slab count, ragged lengths, carried velocity state.  Different summation order,
    hence fp32-noise-level differences, not bitwise equality.  (Until round 4 this test compared the four-wave kernels with the
    eight-wave kernels they had replaced; those were removed in round 5.)"""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    rng = np.random.default_rng(21)
    shapes = ((256, 60), (300, 15), (520, 9))
    lens = {sh: [int(v) for v in rng.integers(1, sh[1] + 1, size=sh[0])] for sh in shapes}
    outs = {}
    for mode in (0, 1):
        with MobilePoserNet.from_numpy(weights, smpl) as n:
            n.set_lstm_mode(mode)
            o = []
            for B, T in shapes:
                L = list(lens[(B, T)])
                L[0] = T
                x = cu(torch_mod, synthetic.make_imu(B, T, seed=B + 3))
                o += [t.clone() for t in n.forward_offline(x, L)]
                o += [t.clone() for t in n.forward_offline(x, L)]
                n.reset_all()
            assert n.device_error() == 0
        outs[mode] = o
    for a, b in zip(outs[0], outs[1]):
        assert float((a - b).abs().max()) < 2e-5


def test_32_slice_fp32_kernel_matches_16_slice(torch_mod, weights, smpl, monkeypatch):
    """mp_lstm_u8 (B <= 32, joints block up to B = 64: 32 slices of 8 units per slab, MFMA tiles of 4 gates x 4 units, K reduction
    and gate transpose in one LDS pass, one XCD per cluster) against the 16-slice kernels (MP_VARIANT slices32=0): one sequence, partly
    filled and full slabs, ragged lengths, carried velocity state, a long sequence, both transports.  Another order of
    summation: equal to fp32 rounding."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    rng = np.random.default_rng(23)
    shapes = ((1, 40), (7, 33), (16, 25), (17, 20), (32, 31), (50, 12), (64, 9), (2, 700))
    lens = {sh: [int(v) for v in rng.integers(1, sh[1] + 1, size=sh[0])] for sh in shapes}
    outs = {}
    for s32, remote in ((0, 0), (1, 0), (1, 1)):
        monkeypatch.setenv("MP_VARIANT", "slices32=%d" % s32)
        with MobilePoserNet.from_numpy(weights, smpl) as n:
            n.set_lstm_mode(1)
            if remote:
                n.set_transport(True)
            o = []
            for B, T in shapes:
                L = list(lens[(B, T)])
                L[0] = T
                x = cu(torch_mod, synthetic.make_imu(B, T, seed=B + 7))
                o += [t.clone() for t in n.forward_offline(x, L)]
                o += [t.clone() for t in n.forward_offline(x, L)]      # carried velocity state
                n.reset_all()
            assert n.device_error() == 0
        outs[(s32, remote)] = o
    for a, b in zip(outs[(0, 0)], outs[(1, 0)]):
        assert float((a - b).abs().max()) < 5e-6
    for a, b in zip(outs[(1, 0)], outs[(1, 1)]):
        assert torch_mod.equal(a, b)                       # the transport never changes a bit


def test_epoch_tagged_exchange_equals_zeroed_exchange(torch_mod, weights, smpl, monkeypatch):
    """The fp32 layer kernels no longer get a zeroed hidden-state exchange area per launch: every launch tags its granules
    with a fresh epoch base (base + step) and the host re-zeroes only before the 32-bit tag would wrap.  Bitwise the same
    outputs as with a memset before every launch (MP_VARIANT epoch_tags=0) -- over many launches on the same areas, different
    shapes sharing a handle, both operand modes in turn (the split-bf16 kernels leave their own words in the area), and
    with the counter started just below the wrap guard so that the re-zeroing path runs several times."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    rng = np.random.default_rng(33)
    shapes = ((256, 40), (40, 25), (300, 9), (1, 60))
    xs = {sh: cu(torch_mod, synthetic.make_imu(sh[0], sh[1], seed=sum(sh))) for sh in shapes}
    lens = {}
    for sh in shapes:
        L = [int(v) for v in rng.integers(1, sh[1] + 1, size=sh[0])]
        L[0] = sh[1]
        lens[sh] = L
    outs = {}
    for tags, start in (("0", None), ("1", None), ("1", "0xEFFFFF00")):
        monkeypatch.setenv("MP_VARIANT", "epoch_tags=%s" % tags + (",epoch_start=%s" % start if start else ""))
        with MobilePoserNet.from_numpy(weights, smpl) as n:
            o = []
            for rep in range(3):
                for mode in (1, 3, 1):
                    n.set_lstm_mode(mode)
                    for sh in shapes:
                        n.reset_all()
                        o += [t.clone() for t in n.forward_offline(xs[sh], lens[sh])]
                        o += [t.clone() for t in n.forward_offline(xs[sh], lens[sh])]
            assert n.device_error() == 0
        outs[(tags, start)] = o
    ref = outs[("0", None)]
    for key in (("1", None), ("1", "0xEFFFFF00")):
        assert len(ref) == len(outs[key])
        for a, b in zip(ref, outs[key]):
            assert torch_mod.equal(a, b), key


def test_g13_evaluate_pose_online_branch_golden(torch_mod, net, monkeypatch):
    """evaluate_pose with ONLINE=1 (evaluate.py:57-65,96-103) on the real network against the reference's own two tables
    for the same sequences (golden G13): forward_offline per sequence, then forward_online frame by frame over the
    sequence padded with five copies of its last frame; velocity state and foot positions carry over between sequences.
    (The fixture's model has run nothing yet, like the reference's when it recorded the tables.)"""
    from mobileposer_amd.data import rotation_matrix_to_r6d
    from mobileposer_amd.evaluate import evaluate_pose
    g = load_golden("g13_evaluate_online.npz")
    ds = [(torch_mod.from_numpy(g[f"s{k}_imu"]), rotation_matrix_to_r6d(torch_mod.from_numpy(g[f"s{k}_pose_t"])).reshape(-1, 144),
           None, torch_mod.from_numpy(g[f"s{k}_tran_t"])) for k in range(int(g["n_seq"]))]
    monkeypatch.setenv("ONLINE", "1")
    out = evaluate_pose(net, ds, verbose=False)
    assert net.device_error() == 0
    np.testing.assert_allclose(npy(out["offline"]), g["offline"], rtol=3e-4, atol=1e-4)
    np.testing.assert_allclose(npy(out["online"]), g["online"], rtol=3e-4, atol=1e-4)

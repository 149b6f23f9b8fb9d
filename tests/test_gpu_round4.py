"""Round-4 GPU parity tests: the inputs the round-3 verdict found missing.

  * weights in the TRAINED regime (synthetic.make_weights(0, profile="trained"): LSTM weights x 3, forget bias + 1, linear1 x 2)
    against golden G14 recorded from the reference -- ragged mixed-combo forward, forward_offline at T = 600, 50 online
    frames -- and against the oracle at the BASELINE size 256 x 125; a second init-scale seed; all 12 sensor combos
    (config.py:60-73, data.py:69-76) in one batch.  Exact-fp32 operands (mode 1, the library default) must meet 1e-4 / 1 mm
    everywhere, and so must the opt-in split-fp16 mode (mode 3) since it was rebuilt on fp16 halves this round.
  * the mesh kernels at the real SMPL size (6890 vertices = 26 x 256 + 234) and at chunk edges (V = 257, 512), golden G15:
    forward_kinematics(calc_mesh), with shape, zero-pose body of a shape, pose blend shapes, the evaluator table.
  * the state a failed call carries forward: after a reported device error with recovery off the next call is finite.
  * the RCCL branch of bench.py on the one GPU there is: a 1-rank torch.distributed.run launch (tests/test_gpu_rccl.py).
"""
import ctypes as C

import numpy as np
import pytest

from conftest import cu, geodesic, load_golden, npy, lstm_test_modes

pytestmark = pytest.mark.gpu

TOL = 1e-4
TOL_TRAN = 1e-3
# What each operand mode is held to (max abs error on raw network outputs / rad on rotations): the north-star bound, 1e-4 / 1 mm,
# on every weight profile, in BOTH modes.  History: until this round mode 3 split its operands into bf16 halves (17 bits) and
# missed the bound on trained-regime weights by 20-200 x (2e-3 on r6d at 64 x 125, 1.6e-2 in 256 x 125: the first run of these
# tests, profiles/r04_parity_errors.txt); it now splits into fp16 halves with weights pre-scaled by 16 (24 bits, mp_lstm_dev.h
# pair_of) and sits at fp32's own noise level (profiles/r04_accuracy.json).  At 256 x 125 on the trained-regime net, where fp32
# implementations differ from each other by more than 1e-4, a mode is held to NOISE_FACTOR x the fp32 oracle's own distance
# from float64 (measured: mode 1 0.7-1.6 x, mode 3 1.8-3.3 x).
MODE_TOL = {"fp32": 1e-4, "x3": 1e-4}
NOISE_FACTOR = {"fp32": 2.0, "x3": 5.0}      # (fp32: 3.0 until round 5; measured 0.7-1.6 x, profiles/r05_accuracy_256x125.json)


def trained_tol(mode):
    return MODE_TOL[mode]


def trained_tol_tran(mode):
    return TOL_TRAN


@pytest.fixture(scope="module")
def weights_trained():
    from mobileposer_amd.synthetic import make_weights
    return make_weights(0, profile="trained")


@pytest.fixture(params=lstm_test_modes())
def tnet(request, torch_mod, weights_trained, smpl):
    """A net with trained-regime weights, once per operand mode."""
    from mobileposer_amd.net import MobilePoserNet
    n = MobilePoserNet.from_numpy(weights_trained, smpl, device="cuda:0")
    n.mode_name = request.param
    n.set_lstm_mode(3 if request.param == "x3" else 1)
    yield n
    n.close()


def test_g14_trained_forward_ragged_mixed_combos(torch_mod, tnet):
    g = load_golden("g14_trained.npz")
    tol = trained_tol(tnet.mode_name)
    lengths = g["lengths"].tolist()
    pose, joints, vel, contact, r6d = tnet.forward(cu(torch_mod, g["imu"]), lengths, return_r6d=True)
    errs = {}
    for b, n in enumerate(lengths):                       # rows past a sequence's length are padding on both sides
        for name, got, want in (("joints", joints, g["joints"]), ("vel", vel, g["vel"]), ("contact", contact, g["contact"]),
                                ("r6d", r6d, g["r6d"])):
            errs[name] = max(errs.get(name, 0.0), float(np.abs(npy(got)[b, :n] - want[b, :n]).max()))
    errs["pose"] = float(geodesic(npy(pose), g["pose"]).max())
    h, c = tnet.velocity.rnn_state
    errs["vel_h"], errs["vel_c"] = float(np.abs(npy(h) - g["vel_h"]).max()), float(np.abs(npy(c) - g["vel_c"]).max())
    print("G14 ragged forward, mode %s: %s" % (tnet.mode_name, {k: "%.2e" % v for k, v in errs.items()}))
    c_err = errs.pop("vel_c")                        # cell state: |c| up to 10 here, held to 1e-4 RELATIVE to its magnitude
    assert max(errs.values()) < tol, errs
    assert c_err < tol * max(1.0, float(np.abs(g["vel_c"]).max())), c_err


def test_g14_trained_offline_600_frames(torch_mod, tnet):
    g = load_golden("g14_trained.npz")
    tol = trained_tol(tnet.mode_name)
    tnet.reset_all()
    pose, joints, tran, contact = tnet.forward_offline(cu(torch_mod, g["off_imu"]), [600])
    e = {"pose": float(geodesic(npy(pose), g["off_pose"]).max()), "joints": float(np.abs(npy(joints) - g["off_joints"]).max()),
         "contact": float(np.abs(npy(contact) - g["off_contact"]).max()), "tran": float(np.abs(npy(tran) - g["off_tran"]).max())}
    print("G14 offline T=600, mode %s: %s" % (tnet.mode_name, {k: "%.2e" % v for k, v in e.items()}))
    assert e["pose"] < tol and e["joints"] < tol and e["contact"] < tol, e
    assert e["tran"] < trained_tol_tran(tnet.mode_name), e     # mode 1: 1 mm after 600 accumulated frames (net.py:154)


def test_g14_trained_online_50_frames(torch_mod, tnet):
    g = load_golden("g14_trained.npz")
    tol = trained_tol(tnet.mode_name)
    tnet.reset_all()
    worst = {"pose": 0.0, "joints": 0.0, "contact": 0.0, "tran": 0.0}
    for k, f in enumerate(g["on_imu"]):
        pose, joints, tran, contact = tnet.forward_online(cu(torch_mod, f))
        worst["pose"] = max(worst["pose"], float(geodesic(npy(pose).reshape(24, 3, 3), g["on_pose"][k].reshape(24, 3, 3)).max()))
        worst["joints"] = max(worst["joints"], float(np.abs(npy(joints)[40] - g["on_joints40"][k]).max()))
        worst["contact"] = max(worst["contact"], float(np.abs(npy(contact) - g["on_contact"][k]).max()))
        worst["tran"] = max(worst["tran"], float(np.abs(npy(tran) - g["on_tran"][k]).max()))
    h, c = tnet.velocity.rnn_state
    worst["vel_c"] = float(np.abs(npy(c) - g["on_vel_c"]).max())
    print("G14 online x50, mode %s: %s" % (tnet.mode_name, {k: "%.2e" % v for k, v in worst.items()}))
    assert worst["pose"] < tol and worst["joints"] < tol and worst["contact"] < tol, worst
    assert worst["vel_c"] < tol * max(1.0, float(np.abs(g["on_vel_c"]).max())), worst
    assert worst["tran"] < trained_tol_tran(tnet.mode_name), worst


@pytest.mark.parametrize("tag", ["tr", "s1"])
@pytest.mark.parametrize("mode", lstm_test_modes())
def test_g14_all_twelve_combos(torch_mod, weights_trained, smpl, tag, mode):
    """Row k of the batch keeps the devices of combo k: all 12 of config.py:60-73 through one forward; trained profile and
    a second init-scale seed (make_weights(1))."""
    from mobileposer_amd.net import MobilePoserNet
    from mobileposer_amd.synthetic import make_weights
    g = load_golden("g14_trained.npz")
    sd = weights_trained if tag == "tr" else make_weights(1)
    with MobilePoserNet.from_numpy(sd, smpl) as n:
        n.set_lstm_mode(3 if mode == "x3" else 1)
        pose, joints, vel, contact, r6d = n.forward(cu(torch_mod, g["c12_imu"]), [40] * 12, return_r6d=True)
        e = {"joints": float(np.abs(npy(joints) - g[f"c12_{tag}_joints"]).max()), "vel": float(np.abs(npy(vel) - g[f"c12_{tag}_vel"]).max()),
             "contact": float(np.abs(npy(contact) - g[f"c12_{tag}_contact"]).max()), "r6d": float(np.abs(npy(r6d) - g[f"c12_{tag}_r6d"]).max())}
        print("G14 12 combos (%s), mode %s: %s" % (tag, mode, {k: "%.2e" % v for k, v in e.items()}))
        assert max(e.values()) < (trained_tol(mode) if tag == "tr" else MODE_TOL[mode]), e
        assert n.device_error() == 0


def _oracle_forward(sd, J, imu, lengths, dtype):
    """The oracle's forward in float32 (what the parity bound is stated against) or with its dtype switched to float64
    (the same arithmetic carried out exactly: the yardstick for "how much of a difference is fp32's own rounding")."""
    from oracle import mp_oracle as O
    O.F32 = dtype
    try:
        ref = O.OracleNet(sd, J)
        pose, joints, vel, contact = ref.forward(imu, lengths)
        return {"pose": np.asarray(pose), "joints": np.asarray(joints), "vel": np.asarray(vel), "contact": np.asarray(contact),
                "r6d": np.asarray(ref._last_r6d)}, float(ref.floor_y)
    finally:
        O.F32 = np.float32


@pytest.mark.parametrize("profile,seed", [("trained", 0), ("init", 1)])
@pytest.mark.parametrize("mode", ["fp32", "x3"])
def test_baseline_size_vs_oracle_other_weights(torch_mod, smpl, profile, seed, mode):
    """256 x 125 (the BASELINE shape, every schedule decision of the headline) against the oracle with weights other than
    the one draw rounds 1-3 tested: the trained profile, and a second init-scale seed.  All 12 combos appear in the batch
    (row b uses combo b % 12).

    Bound: 1e-4 against the fp32 oracle -- or, where fp32 itself does not resolve 1e-4 on this net, fp32's own noise: the
    trained-regime net amplifies rounding so much that two fp32 implementations differ by more than 1e-4 somewhere in 32 000
    frames (numpy fp32 vs the same arithmetic in float64: 1.6e-4 on the contact logits already at 64 x 125,
    profiles/r04_accuracy.json).  There the library must be as close to the float64 result as the fp32 oracle is (x 3)."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.config import amass
    from mobileposer_amd.net import MobilePoserNet
    from oracle import mp_oracle as O
    B, T = 256, 125
    names = list(amass.combos)
    sd = synthetic.make_weights(seed, profile=profile)
    imu = synthetic.make_imu(B, T, seed=40 + seed, combo=[names[b % 12] for b in range(B)])
    ref, floor_y = _oracle_forward(sd, smpl["J"], imu, [T] * B, np.float32)
    truth, _ = _oracle_forward(sd, smpl["J"], imu, [T] * B, np.float64) if profile == "trained" else (None, None)
    with MobilePoserNet.from_numpy(sd, smpl) as n:
        n.set_lstm_mode(3 if mode == "x3" else 1)
        pose, joints, vel, contact, r6d = n.forward(cu(torch_mod, imu), [T] * B, return_r6d=True)
        got = {"joints": npy(joints), "vel": npy(vel), "contact": npy(contact), "r6d": npy(r6d), "pose": npy(pose)}
        err = lambda a, b, k: float(geodesic(a, b).max()) if k == "pose" else float(np.abs(np.asarray(a, np.float64) - b).max())
        e = {k: err(got[k], ref[k], k) for k in got}
        print("256x125 vs fp32 oracle, %s seed %d, mode %s: %s" % (profile, seed, mode, {k: "%.2e" % v for k, v in e.items()}))
        tol = trained_tol(mode) if profile == "trained" else MODE_TOL[mode]
        if truth is None:
            assert max(e.values()) < tol, e
        else:
            e64 = {k: err(got[k], truth[k], k) for k in got}                  # library vs exact arithmetic
            n64 = {k: err(ref[k], truth[k], k) for k in got}                  # fp32 oracle vs exact arithmetic
            print("   vs float64: library %s | fp32 oracle %s" % ({k: "%.2e" % v for k, v in e64.items()}, {k: "%.2e" % v for k, v in n64.items()}))
            for k in got:
                assert e[k] < tol or e64[k] < NOISE_FACTOR[mode] * n64[k], (k, e[k], e64[k], n64[k])
        # translation of every 16th row through the batched solver, 1 mm
        tran = torch_mod.empty(B, T, 3, device="cuda")
        n.translate_offline_into(joints, vel.reshape(B, T, 72), contact, (C.c_int32 * B)(*([T] * B)), tran)
        tran_h = npy(tran)
        for b in range(0, B, 16):
            rt = O.translate_offline(ref["joints"][b].reshape(T, 24, 3), ref["vel"][b], ref["contact"][b], floor_y)
            assert np.abs(tran_h[b] - rt).max() < (trained_tol_tran(mode) if profile == "trained" else TOL_TRAN), b
        assert n.device_error() == 0 and n.recovery_count == 0


# ---- the mesh kernels at the real size -------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big_body(torch_mod):
    from mobileposer_amd.body_model import ParametricModel
    from mobileposer_amd.synthetic import synthetic_smpl
    data = synthetic_smpl(n_vertex=6890)
    bm = ParametricModel(data=data)
    yield bm, data
    bm.close()


def test_g15_mesh_6890_golden(torch_mod, big_body):
    """forward_kinematics(calc_mesh=True) at V = 6890 (27 vertex chunks, the last one 234 wide), with and without shape,
    against the reference's own outputs; and get_zero_pose_joint_and_vertex(shape) (articulate/model.py:84-89)."""
    bm, _ = big_body
    g = load_golden("g15_mesh6890.npz")
    pose, tran, shape = cu(torch_mod, g["pose"]), cu(torch_mod, g["tran"]), cu(torch_mod, g["shape"])
    _, jg, vg = bm.forward_kinematics(pose, tran=tran, calc_mesh=True)
    assert tuple(vg.shape) == (3, 6890, 3)
    assert np.abs(npy(jg) - g["joint"]).max() < 1e-5 and np.abs(npy(vg) - g["vert"]).max() < 1e-5
    _, jg, vg = bm.forward_kinematics(pose, shape=shape, tran=tran, calc_mesh=True)
    assert np.abs(npy(jg) - g["shape_joint"]).max() < 1e-5 and np.abs(npy(vg) - g["shape_vert"]).max() < 2e-5
    j0, v0 = bm.get_zero_pose_joint_and_vertex(shape[:2])
    assert tuple(j0.shape) == (2, 24, 3) and tuple(v0.shape) == (2, 6890, 3)
    assert np.abs(npy(j0) - g["zero_joint"]).max() < 1e-5 and np.abs(npy(v0) - g["zero_vert"]).max() < 1e-5
    jn, vn = bm.get_zero_pose_joint_and_vertex()                       # shape None: host constants, as before
    assert jn.shape == (24, 3) and vn.shape == (6890, 3) and np.abs(jn[0]).max() == 0


def test_g15_pose_blend_shapes(torch_mod):
    """ParametricModel(use_pose_blendshape=True) (articulate/model.py:236-238): vertices skinned from
    v + posedirs . (pose[1:] - I), with and without a shape, against the reference."""
    from mobileposer_amd.body_model import ParametricModel
    from mobileposer_amd.synthetic import synthetic_smpl
    g = load_golden("g15_mesh6890.npz")
    bm = ParametricModel(data=synthetic_smpl(n_vertex=6890), use_pose_blendshape=True)
    try:
        pose, tran, shape = cu(torch_mod, g["pose"]), cu(torch_mod, g["tran"]), cu(torch_mod, g["shape"])
        _, jg, vg = bm.forward_kinematics(pose, shape=shape[:1], tran=tran, calc_mesh=True)
        assert np.abs(npy(jg) - g["blend_joint"]).max() < 1e-5
        assert np.abs(npy(vg) - g["blend_vert"]).max() < 2e-5
        _, _, vg = bm.forward_kinematics(pose, calc_mesh=True)
        assert np.abs(npy(vg) - g["blend_vert_noshape"]).max() < 2e-5
    finally:
        bm.close()


@pytest.mark.parametrize("V", [96, 257, 512, 6890])
def test_mesh_and_evaluator_vs_oracle_at_chunk_edges(torch_mod, weights, V):
    """mp_lbs / mp_shape_* / mp_eval_metrics for N = 40 frames at V = 96 (one partial chunk), 257 (a 1-vertex tail), 512 (two
    full chunks) and 6890 (the real mesh) against the numpy oracle (size-agnostic; pinned at V = 96 by G6 / G9 / G12 and at
    V = 6890 by G15)."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    from oracle import mp_oracle as O
    data = synthetic.synthetic_smpl(n_vertex=V)
    rng = np.random.Generator(np.random.PCG64(500 + V))
    N = 40
    pose_t = synthetic._random_rotations(rng, N * 24).reshape(N, 24, 3, 3).astype(np.float32)
    from scipy.spatial.transform import Rotation
    noise = Rotation.from_rotvec((rng.standard_normal((N * 24, 3)) * 0.1)).as_matrix().reshape(N, 24, 3, 3)
    pose_p = np.einsum("njab,njbc->njac", pose_t, noise).astype(np.float32)
    tran_t = np.cumsum(rng.standard_normal((N, 3)) * 0.02, axis=0).astype(np.float32)
    tran_p = (tran_t + np.cumsum(rng.standard_normal((N, 3)) * 0.005, axis=0)).astype(np.float32)
    shape = (rng.standard_normal((N, 10)) * 1.2).astype(np.float32)
    with MobilePoserNet.from_numpy(weights, data) as n:
        Rg, jg, vg = n.forward_kinematics(cu(torch_mod, pose_p), tran=cu(torch_mod, tran_p), calc_mesh=True)
        rRg, rjg, rvg = O.forward_kinematics_mesh(pose_p, data, tran=tran_p)
        assert np.abs(npy(vg) - rvg).max() < 2e-5 and np.abs(npy(jg) - rjg).max() < 1e-5
        Rg, jg, vg = n.forward_kinematics(cu(torch_mod, pose_p), tran=cu(torch_mod, tran_p), calc_mesh=True, shape=cu(torch_mod, shape))
        rRg, rjg, rvg = O.forward_kinematics_shape(pose_p, data, shape, tran=tran_p)
        assert np.abs(npy(vg) - rvg).max() < 3e-5 and np.abs(npy(jg) - rjg).max() < 2e-5
        ign = O.IGNORED
        table = npy(n.eval_metrics(pose_p, pose_t, tran_p, tran_t, fps=30, joint_mask=[2, 5, 16, 20], ignored=ign))
        pp, pt = pose_p.copy(), pose_t.copy()
        pp[:, ign] = np.eye(3)
        pt[:, ign] = np.eye(3)
        want = O.full_motion_evaluator(pp, pt, data, tran_p, tran_t, fps=30)
        np.testing.assert_allclose(table, want, rtol=2e-4, atol=2e-5)
        with pytest.raises(ValueError):
            n.eval_metrics(pose_p, pose_t, joint_mask=[32])
        assert n.device_error() == 0


# ---- carried state after a reported device error ---------------------------------------------------------
def _starve(m, skip=0, launches=1):
    assert m._lib.mp_debug_drop_workgroup(m._h, 8, skip, launches) == 0


def test_call_after_a_reported_error_is_finite_without_reset(torch_mod, weights, smpl, monkeypatch):
    """Recovery off: a starved call leaves NaN in the velocity state it updates in place.  Once the error has been reported
    (finish() raises) the handle drops that state, so the NEXT call -- with no reset by the caller -- is finite and equals
    a call from zero state."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    monkeypatch.setenv("MP_WAIT_MS", "15")
    B, T = 256, 20
    x = cu(torch_mod, synthetic.make_imu(B, T, seed=80))
    with MobilePoserNet.from_numpy(weights, smpl) as m:
        m.set_lstm_mode(1)
        m.set_recovery(False)
        want = [t.clone() for t in m.forward_offline(x, [T] * B)]          # from zero state
        m.finish()
        m.reset_all()
        _starve(m, skip=4)                                                  # velocity layer 0 of the next forward
        m.forward_offline(x, [T] * B)
        with pytest.raises(RuntimeError, match="state it carried forward is lost"):
            m.finish()
        assert m.velocity.rnn_state is None                                # dropped, like `model.velocity.rnn_state = None`
        got = m.forward_offline(x, [T] * B)                                 # NO reset in between
        m.finish()
        for a, b in zip(want, got):
            assert bool(torch_mod.isfinite(b).all())
            assert float((a - b).abs().max()) < 2e-5


def test_stream_tick_after_a_reported_error_is_finite(torch_mod, weights, smpl, monkeypatch):
    """The same for streaming: a starved tick with recovery off derives root height / position and last foot positions from
    NaN outputs.  After the error is reported every stream is back to its state after construction; later ticks are finite
    and equal those of a fresh model fed the same frames."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    monkeypatch.setenv("MP_WAIT_MS", "15")
    S = 256
    frames = cu(torch_mod, synthetic.make_imu(S, 6, seed=81))
    with MobilePoserNet.from_numpy(weights, smpl) as m, MobilePoserNet.from_numpy(weights, smpl) as fresh:
        for n in (m, fresh):
            n.set_lstm_mode(1)
            n.stream_create(S)
        m.set_recovery(False)
        m.stream_step(frames[:, 0])
        m.finish()
        _starve(m, skip=4)
        m.stream_step(frames[:, 1])                                         # poisoned tick
        with pytest.raises(RuntimeError, match="all streams reset"):
            m.finish()
        for k in (2, 3, 4):
            got = m.stream_step(frames[:, k])
            want = fresh.stream_step(frames[:, k])
            m.finish()
            for a, b in zip(want, got):
                assert bool(torch_mod.isfinite(b).all()), k
                assert float((a - b).abs().max()) < 2e-5, k


def test_rnn_forward_repairs_an_aliased_state(torch_mod, weights, smpl, monkeypatch):
    """mp_rnn_forward with state_in_dev == state_out_dev (in-place state): a starved fused run overwrites the initial state
    with NaN; recovery restores it from its snapshot before the per-step re-run."""
    import warnings
    from mobileposer_amd.net import MobilePoserNet, _ptr
    monkeypatch.setenv("MP_WAIT_MS", "15")
    B, T = 256, 12
    rng = np.random.Generator(np.random.PCG64(82))
    x = cu(torch_mod, (rng.standard_normal((B, T, 132)) * 0.5).astype(np.float32))
    st0 = cu(torch_mod, (rng.standard_normal((2, 2, B, 256)) * 0.3).astype(np.float32))
    lens = (C.c_int32 * B)(*([T] * B))
    with MobilePoserNet.from_numpy(weights, smpl) as m:
        m.set_lstm_mode(1)
        y_ref = torch_mod.empty(B, T, 72, device="cuda")
        st_ref = st0.clone()
        assert m._lib.mp_rnn_forward(m._h, 3, _ptr(x), lens, B, T, _ptr(y_ref), _ptr(st_ref), _ptr(st_ref), m._stream()) == 0
        y = torch_mod.empty(B, T, 72, device="cuda")
        st = st0.clone()
        _starve(m)
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            assert m._lib.mp_rnn_forward(m._h, 3, _ptr(x), lens, B, T, _ptr(y), _ptr(st), _ptr(st), m._stream()) == 0
        assert m.recovery_count == 1
        assert bool(torch_mod.isfinite(y).all()) and bool(torch_mod.isfinite(st).all())
        assert float((y - y_ref).abs().max()) < 2e-5 and float((st - st_ref).abs().max()) < 2e-5


# ---- coalesced kinematics kernels (round 4) ---------------------------------------------------------------------------
@pytest.mark.parametrize("N", [1, 7, 8, 13, 32000])
def test_coalesced_fk_and_ik_equal_the_scalar_kernels_bitwise(torch_mod, net, N):
    """mp_fk_lds / mp_r6d_ik_lds (16-byte pieces through LDS, chosen for 16-byte aligned buffers) against mp_fk / mp_r6d_ik
    (scalar accesses, what a misaligned buffer gets): same arithmetic in the same order, so every bit agrees; ragged last
    workgroup (N not a multiple of 8); translation added; and both against the oracle."""
    from oracle import mp_oracle as O
    rng = np.random.Generator(np.random.PCG64(600 + N))
    r6d = rng.standard_normal((N, 96)).astype(np.float32)
    tran = rng.standard_normal((N, 3)).astype(np.float32)
    pad = lambda a: torch_mod.cat((torch_mod.zeros(1, device="cuda"), cu(torch_mod, a).reshape(-1)))[1:].reshape(a.shape)   # 4-byte aligned only
    pose_a = net._reduced_global_to_full(cu(torch_mod, r6d))
    pose_u = net._reduced_global_to_full(pad(r6d))
    assert pad(r6d).data_ptr() % 16 != 0 and torch_mod.equal(pose_a, pose_u)
    Rg_a, jg_a = net.forward_kinematics(pose_a, tran=cu(torch_mod, tran))
    Rg_u, jg_u = net.forward_kinematics(pad(npy(pose_a)), tran=cu(torch_mod, tran))
    assert torch_mod.equal(Rg_a, Rg_u) and torch_mod.equal(jg_a, jg_u)
    if N <= 13:
        ref = O.reduced_global_to_full(r6d)
        assert np.abs(npy(pose_a) - ref).max() < 1e-5
        rRg, rjg = O.forward_kinematics(ref, net.bodymodel.J, tran=tran)
        assert np.abs(npy(Rg_a) - rRg).max() < 1e-5 and np.abs(npy(jg_a) - rjg).max() < 1e-5


@pytest.mark.parametrize("B,T", [(257, 100), (200, 30)])
def test_linear_layer_kernels_on_ragged_row_counts(torch_mod, net, weights, smpl, B, T):
    """The full-batch linear-layer kernels of round 4 (mp_gemm_l2l1, mp_gemm_f32_wide: one wave = 32 rows x all columns) when
    B * T is NOT a multiple of 32 -- the instantiations with per-row bounds checks, whose last wave owns a partial tile:
    257 x 100 = 25 700 rows (wide + l2l1), 200 x 30 = 6 000 (l2l1 only).  Against the oracle on a subset of the sequences
    (sequences never interact), the last ones included, and on every row for NaN / garbage past the valid rows."""
    from mobileposer_amd import synthetic
    from oracle import mp_oracle as O
    imu = synthetic.make_imu(B, T, seed=B + T)
    net.reset_all()
    pose, joints, vel, contact = net.forward(cu(torch_mod, imu), [T] * B)
    for t in (pose, joints, vel, contact):
        assert bool(torch_mod.isfinite(t).all())
    rows = [0, 1, B // 2, B - 2, B - 1]
    ref = O.OracleNet(weights, smpl["J"])
    rpose, rjoints, rvel, rcontact = ref.forward(imu[rows], [T] * len(rows))
    assert np.abs(npy(joints)[rows] - rjoints).max() < TOL
    assert np.abs(npy(vel).reshape(B, T, 72)[rows] - rvel).max() < TOL
    assert np.abs(npy(contact)[rows] - rcontact).max() < TOL
    assert geodesic(npy(pose).reshape(B, T, 24, 3, 3)[rows].reshape(-1, 24, 3, 3), rpose).max() < TOL
    assert net.device_error() == 0


def test_one_step_call_on_a_carried_state_vs_oracle(torch_mod, weights_trained, smpl):
    """T = 1 with a carried velocity state: every fused layer launch is ONE step whose recurrent operand is the state the
    previous call left in place.  (Until round 4 a workgroup read that operand for all 16 rows of its slab from the state
    buffer while a faster workgroup of the same cluster could already be writing its final state there; the initial state now
    travels through the exchange like every other step's.)  Five calls against the oracle, saturated-gate weights."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    from oracle import mp_oracle as O
    B = 256
    ref = O.OracleNet(weights_trained, smpl["J"])
    with MobilePoserNet.from_numpy(weights_trained, smpl) as n:
        n.set_lstm_mode(1)
        for call in range(5):
            imu = synthetic.make_imu(B, 1, seed=900 + call)
            pose, joints, vel, contact = n.forward(cu(torch_mod, imu), [1] * B)
            rpose, rjoints, rvel, rcontact = ref.forward(imu, [1] * B)
            assert np.abs(npy(joints) - rjoints).max() < 1e-4, call
            assert np.abs(npy(vel).reshape(rvel.shape) - rvel).max() < 1e-4, call
            assert np.abs(npy(contact) - rcontact).max() < 1e-4, call
        assert n.device_error() == 0


def test_nan_sample_poisons_its_sequence_and_only_it(torch_mod, net, weights, smpl):
    """One NaN in one IMU sample: the reference (torch: relu(NaN) = NaN, NaN gates) returns NaN joints, velocity and contact
    for EVERY frame of that sequence (the bidirectional joints layers carry it both ways, the other blocks read the joints) and the
    NaN -> 0 rule of r6d_to_rotation_matrix (angular.py:181) keeps the pose finite; every other sequence is untouched."""
    from mobileposer_amd import synthetic
    B, T, b, t = 40, 30, 17, 11
    imu = synthetic.make_imu(B, T, seed=321)
    net.reset_all()
    clean = [x.clone() for x in net.forward(cu(torch_mod, imu), [T] * B)]
    bad = imu.copy()
    bad[b, t, 7] = np.nan
    net.reset_all()
    got = [x.clone() for x in net.forward(cu(torch_mod, bad), [T] * B)]
    others = [i for i in range(B) if i != b]
    pose = got[0].reshape(B, T, -1)
    assert bool(torch_mod.isfinite(pose).all())
    for k in (1, 2, 3):                                        # joints: all frames; velocity and contact read them
        assert bool(torch_mod.isnan(got[k][b]).all()), k
    for k in (1, 2, 3):
        assert torch_mod.equal(got[k][others], clean[k][others])
    assert torch_mod.equal(pose[others], clean[0].reshape(B, T, -1)[others])
    assert net.device_error() == 0


@pytest.mark.parametrize("B", [48, 256])
def test_tagged_exchange_over_short_launches_vs_oracle(torch_mod, weights_trained, smpl, B):
    """The tagged hidden-state exchange starts every launch with the tags the previous one did NOT leave behind (two bits per
    exchange area, kept by the host: LstmPersistArgs::tag_flip).  How many times a parity slot is written depends on T, so
    launches of T = 2, 3, 4, 7, 1, 6 alternate on ONE handle -- every (B, T) has its own areas, revisited three times each --
    with the velocity state carried from call to call, against the oracle.  A wrong tag would either stall (device error) or
    accept a word of the previous launch (wrong numbers)."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    from oracle import mp_oracle as O
    ref = O.OracleNet(weights_trained, smpl["J"])
    with MobilePoserNet.from_numpy(weights_trained, smpl) as n:
        n.set_lstm_mode(1)
        for rnd in range(3):
            for T in (2, 3, 4, 7, 1, 6):
                imu = synthetic.make_imu(B, T, seed=1000 + 10 * rnd + T)
                L = [T] * B
                L[B // 3] = max(1, T - 1)
                pose, joints, vel, contact = n.forward(cu(torch_mod, imu), L)
                rpose, rjoints, rvel, rcontact = ref.forward(imu, L)
                assert np.abs(npy(joints) - rjoints).max() < 1e-4, (rnd, T)
                assert np.abs(npy(vel).reshape(rvel.shape) - rvel).max() < 2e-4, (rnd, T)
                assert np.abs(npy(contact) - rcontact).max() < 2e-4, (rnd, T)
        assert n.device_error() == 0 and n.recovery_count == 0


def test_g16_nan_sample_golden(torch_mod, net):
    """G16a, recorded from the reference: a NaN in one IMU sample -> that sequence's joints / velocity / contact NaN in every
    frame, its pose all-zero matrices (angular.py:181), the other four sequences as if nothing had happened."""
    g = load_golden("g16_corners.npz")
    net.reset_all()
    pose, joints, vel, contact = net.forward(cu(torch_mod, g["nan_imu"]), [24] * 5)
    for got, key in ((joints, "nan_joints"), (vel, "nan_vel"), (contact, "nan_contact")):
        a, b = npy(got).reshape(g[key].shape), g[key]
        assert (np.isnan(a) == np.isnan(b)).all(), key
        m = ~np.isnan(b)
        assert np.abs(a[m] - b[m]).max() < 1e-4, key
    assert not np.isnan(npy(pose)).any() and np.abs(npy(pose).reshape(g["nan_pose"].shape) - g["nan_pose"]).max() < 1e-4
    assert net.device_error() == 0


def test_g16_one_frame_calls_golden(torch_mod, weights_trained, smpl):
    """G16b, recorded from the reference: five forward() calls of one frame each on a carried velocity state (B = 4: the
    32-slice kernels) -- and the same five calls with the four sequences repeated 64 times (B = 256: the tagged-exchange
    kernels, every fused launch a single step)."""
    from mobileposer_amd.net import MobilePoserNet
    g = load_golden("g16_corners.npz")
    for rep in (1, 64):
        with MobilePoserNet.from_numpy(weights_trained, smpl) as n:
            n.set_lstm_mode(1)
            for k in range(5):
                x = np.tile(g["one_imu"][:, k:k + 1], (rep, 1, 1))
                pose, joints, vel, contact = n.forward(cu(torch_mod, x), [1] * (4 * rep))
                for got, key in ((joints, f"one_joints{k}"), (vel, f"one_vel{k}"), (contact, f"one_contact{k}")):
                    want = np.tile(g[key].reshape(4, -1), (rep, 1))
                    assert np.abs(npy(got).reshape(4 * rep, -1) - want).max() < 1e-4, (rep, k, key)
            h, c = n.velocity.rnn_state
            assert np.abs(npy(h) - np.tile(g["one_vel_h"], (1, rep, 1))).max() < 1e-4
            assert n.device_error() == 0


def test_initial_state_no_lstm_produces_is_taken_by_the_per_step_kernels(torch_mod, weights, smpl):
    """nn.LSTM accepts ANY (h0, c0).  The fused kernels exchange hidden states with a tag in bit 30 of the word, which is free
    only for |h| < 2 -- every state an LSTM produced -- so an initial |h| >= 2 (or a NaN) is detected at launch (device code
    2000000) and the call repaired by the per-step kernels (recovery on, the default): the result matches the oracle, one
    recovery is counted, and the next ordinary call runs on the fused kernels again."""
    import warnings
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    from oracle import mp_oracle as O
    B, T = 256, 6
    rng = np.random.Generator(np.random.PCG64(404))
    x = (rng.standard_normal((B, T, 132)) * 0.5).astype(np.float32)
    h0 = (rng.standard_normal((2, B, 256)) * 0.3).astype(np.float32)
    c0 = (rng.standard_normal((2, B, 256)) * 0.5).astype(np.float32)
    h0[0, 17, 5] = 3.25                                       # no LSTM output
    h0[1, 200, 77] = -2.0
    with MobilePoserNet.from_numpy(weights, smpl) as n:
        n.set_lstm_mode(1)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            y, (h, c) = n.rnn_forward("velocity", cu(torch_mod, x), [T] * B, (cu(torch_mod, h0), cu(torch_mod, c0)))
        assert n.recovery_count == 1 and any("2000000" in str(i.message) for i in w), [str(i.message) for i in w]
        ry, (rh, rc) = O.rnn_forward(weights, O.PREFIX["velocity"], x, [T] * B, (h0, c0))
        assert np.abs(npy(y) - ry).max() < 1e-4 and np.abs(npy(h) - rh).max() < 1e-4 and np.abs(npy(c) - rc).max() < 1e-4
        h0[0, 17, 5], h0[1, 200, 77] = 0.25, -0.5
        y, (h, c) = n.rnn_forward("velocity", cu(torch_mod, x), [T] * B, (cu(torch_mod, h0), cu(torch_mod, c0)))
        ry, (rh, rc) = O.rnn_forward(weights, O.PREFIX["velocity"], x, [T] * B, (h0, c0))
        assert np.abs(npy(y) - ry).max() < 1e-4 and np.abs(npy(h) - rh).max() < 1e-4
        assert n.recovery_count == 1 and n.device_error() == 0

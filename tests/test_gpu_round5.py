"""Round-5 GPU tests: what the round-4 verdict found unverified.

  * the call bench.py times -- mp_forward_offline WITH the FK outputs requested -- against the oracle at 256 x 125 (every
    row) and 1024 x 125 (sampled rows), two consecutive calls (carried velocity state), all seven outputs
    (models/net.py:121-154, articulate/model.py:208-232);
  * a NaN sample no longer puts a handle on the slow path: S = 512 streams with one glitching sensor, and three forward
    calls at 256 x 125 (reference behaviour: velocity.py:45-48 keeps the NaN state, angular.py:181 zeroes the pose);
  * device index handling (mp_create on an index the process cannot see; a HIP_VISIBLE_DEVICES-remapped index 0);
  * the library's build id equals the md5 of the sources beside it;
  * the fused IK + FK kernel of the forward's tail against the two stand-alone kernels, bitwise;
  * mp_stream_replay (N forward_online calls as one call: evaluate.py:62-64) against goldens G5 / G14, split in two, continued by
    single ticks, on the per-step kernels, after a starved launch, and against the calls one by one (values and wall time).
"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import REPO, cu, geodesic, npy

pytestmark = pytest.mark.gpu

TOL = 1e-4
TOL_TRAN = 1e-3


def _offline_buffers(torch, B, T):
    f32 = torch.float32
    mk = lambda *shape: torch.empty(*shape, device="cuda", dtype=f32)
    return {"pose": mk(B * T, 24, 3, 3), "joints": mk(B, T, 72), "vel": mk(B, T, 72), "contact": mk(B, T, 2),
            "tran": mk(B, T, 3), "rglob": mk(B * T, 24, 3, 3), "jglob": mk(B * T, 24, 3)}


@pytest.mark.parametrize("B,rows", [(256, None), (1024, 32)])
def test_the_call_bench_times_vs_oracle(torch_mod, net, weights, smpl, B, rows):
    """mp_forward_offline(..., rglobal_dev, joint_dev): forward + FK + solver in one call, twice in a row (the second call
    starts from the velocity state the first one left, Q1).  Every output against the oracle: all rows at 256 x 125, `rows`
    sampled sequences at 1024 x 125 (sequences never interact, so the oracle runs on those rows alone)."""
    from mobileposer_amd import synthetic
    from oracle import mp_oracle as O
    T = 125
    imu = synthetic.make_imu(B, T, seed=1)
    pick = np.arange(B) if rows is None else np.sort(np.random.Generator(np.random.PCG64(5)).choice(B, rows, replace=False))
    ref = O.OracleNet(weights, smpl["J"])
    o = _offline_buffers(torch_mod, B, T)
    lens = (C.c_int32 * B)(*([T] * B))
    x = cu(torch_mod, imu)
    net.reset_all()
    for call in range(2):
        for t in o.values():
            t.fill_(float("nan"))                      # nothing may survive from the previous call
        net.forward_offline_into(x, lens, o["pose"], o["joints"], o["vel"], o["contact"], o["tran"], o["rglob"], o["jglob"])
        rpose, rjoints, rvel, rcontact = ref.forward(imu[pick], [T] * len(pick))
        rRg, rjg = O.forward_kinematics(rpose, smpl["J"])
        n = len(pick)
        got = {k: npy(v) for k, v in o.items()}
        assert np.abs(got["joints"][pick] - rjoints).max() < TOL, call
        assert np.abs(got["vel"][pick] - rvel.reshape(n, T, 72)).max() < TOL, call
        assert np.abs(got["contact"][pick] - rcontact).max() < TOL, call
        assert geodesic(got["pose"].reshape(B, T, 24, 3, 3)[pick], rpose.reshape(n, T, 24, 3, 3)).max() < TOL, call
        assert geodesic(got["rglob"].reshape(B, T, 24, 3, 3)[pick], rRg.reshape(n, T, 24, 3, 3)).max() < TOL, call
        assert np.abs(got["jglob"].reshape(B, T, 24, 3)[pick] - rjg.reshape(n, T, 24, 3)).max() < TOL, call
        for i, b in enumerate(pick):
            rt = O.translate_offline(rjoints[i].reshape(T, 24, 3), rvel.reshape(n, T, 72)[i], rcontact[i], ref.floor_y)
            assert np.abs(got["tran"][b] - rt).max() < TOL_TRAN, (call, b)
        assert all(np.isfinite(v).all() for v in got.values()), call
    assert net.device_error() == 0 and net.recovery_count == 0


def test_one_glitching_sensor_among_512_streams(torch_mod, weights, smpl):
    """configs[4] with a NaN frame into stream 137 at tick 5, 60 ticks.  The reference keeps the NaN in that stream's velocity
    LSTM state for good (velocity.py:45-48; reset() does not clear it, Q1).  Round 4 answered every later tick with a device
    code, a re-run on the per-step kernels and a warning -- for all 512 streams.  Now: the other 511 streams are bitwise what
    they are in a clean run, stream 137 is NaN exactly where the oracle's forward_online is, no recovery, no warning, and a
    tick costs what a clean tick costs."""
    import warnings
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    from oracle import mp_oracle as O
    S, n, bad_s, bad_k = 512, 60, 137, 5
    frames = synthetic.make_imu(S, n, seed=91)
    bad = frames.copy()
    bad[bad_s, bad_k, 13] = np.nan

    def run(fr):
        outs, dts = [], []
        with MobilePoserNet.from_numpy(weights, smpl) as net:
            net.set_lstm_mode(1)
            net.stream_create(S)
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                for k in range(n):
                    xk = cu(torch_mod, fr[:, k])
                    torch_mod.cuda.synchronize()
                    t0 = time.perf_counter()
                    o = net.stream_step(xk)
                    torch_mod.cuda.synchronize()
                    if k >= 10:
                        dts.append(time.perf_counter() - t0)
                    outs.append([npy(t) for t in o])
            assert net.recovery_count == 0 and not w, [str(i.message) for i in w]
            assert net.device_error() == 0
        # (the MEDIAN tick: every tick here follows a pause -- the copies of its results to the host -- and so runs on the restarting
        #  loaded clock of profiles/r06_idle_probe_S512.txt, 3.0-3.2 ms instead of 2.75; the mean of 50 such ticks of two runs
        #  differed by more than 15 % in one of three suite runs of round 6's validation)
        return outs, float(np.median(dts))

    clean, t_clean = run(frames)
    got, t_bad = run(bad)
    others = np.array([s for s in range(S) if s != bad_s])
    for k in range(n):
        for a, b in zip(clean[k], got[k]):
            assert np.array_equal(a[others], b[others]), k
    ref = O.OracleNet(weights, smpl["J"])
    with np.errstate(all="ignore"):
        for k in range(n):
            rp, rj, rr, rc = ref.forward_online(bad[bad_s, k])
            pose, joints, root, contact = (t[bad_s] for t in got[k])
            for name, g, r in (("pose", pose.reshape(-1), rp.reshape(-1)), ("joints", joints, rj), ("root", root, rr), ("contact", contact, rc)):
                assert np.array_equal(np.isnan(g), np.isnan(r)), (k, name)
                ok = ~np.isnan(r)
                assert np.abs(g[ok] - r[ok]).max(initial=0.0) < (TOL_TRAN if name == "root" else TOL), (k, name)
    print("tick: clean %.3f ms, with one NaN stream %.3f ms" % (1e3 * t_clean, 1e3 * t_bad))
    assert t_bad < 1.2 * t_clean + 2e-5         # (a slow path for the NaN stream would be 2 x and more; 20 %: box noise between two runs)


def test_nan_sample_then_three_forwards_at_baseline_size(torch_mod, net, weights, smpl):
    """forward x 3 at 256 x 125 with one NaN sample in sequence 77 of the FIRST call: calls 2 and 3 start from a velocity state
    whose row 77 is NaN (the reference: same).  Sequence 77's velocity stays NaN, its other outputs recover, every other
    sequence is bitwise what a clean run gives, and no call takes the recovery path."""
    from mobileposer_amd import synthetic
    from oracle import mp_oracle as O
    B, T, b = 256, 125, 77
    imu = synthetic.make_imu(B, T, seed=17)
    bad = imu.copy()
    bad[b, 60, 3] = np.nan
    lengths = [T] * B
    net.reset_all()
    clean = [[t.clone() for t in net.forward(cu(torch_mod, imu), lengths)] for _ in range(3)]
    h_clean, c_clean = net.velocity.rnn_state
    net.reset_all()
    got = [[t.clone() for t in net.forward(cu(torch_mod, bad if k == 0 else imu), lengths)] for k in range(3)]
    h_got, c_got = net.velocity.rnn_state
    others = [i for i in range(B) if i != b]
    for k in range(3):
        for i, (a, g) in enumerate(zip(clean[k], got[k])):
            a, g = a.reshape(B, T, -1), g.reshape(B, T, -1)
            assert torch_mod.equal(a[others], g[others]), (k, i)
        assert bool(torch_mod.isnan(got[k][2].reshape(B, T, 72)[b]).all()), k          # velocity of sequence 77: NaN for good
        if k > 0:                                                                        # joints / pose / contact do not read it
            for i in (0, 1, 3):
                assert torch_mod.equal(clean[k][i].reshape(B, T, -1)[b], got[k][i].reshape(B, T, -1)[b]), (k, i)
    assert torch_mod.equal(h_clean[:, others], h_got[:, others]) and torch_mod.equal(c_clean[:, others], c_got[:, others])
    assert bool(torch_mod.isnan(h_got[:, b]).all()) and bool(torch_mod.isnan(c_got[:, b]).all())
    # the oracle agrees on where the NaNs are (rows: sequence 77 and one neighbour)
    ref = O.OracleNet(weights, smpl["J"])
    with np.errstate(all="ignore"):
        for k in range(3):
            x = (bad if k == 0 else imu)[[b, b + 1]]
            rpose, rjoints, rvel, rcontact = ref.forward(x, [T, T])
            for name, g, r in (("joints", got[k][1], rjoints), ("vel", got[k][2].reshape(B, T, 72), rvel.reshape(2, T, 72)),
                               ("contact", got[k][3], rcontact)):
                g = npy(g)[[b, b + 1]]
                assert np.array_equal(np.isnan(g), np.isnan(r)), (k, name)
                ok = ~np.isnan(r)
                assert np.abs(g[ok] - r[ok]).max(initial=0.0) < TOL, (k, name)
    assert net.device_error() == 0 and net.recovery_count == 0


def test_state_code_does_not_switch_the_placement_tables_off(torch_mod, weights, smpl):
    """An initial |h| >= 2 is a property of the caller's state, not of the placement: the call is repaired by the per-step
    kernels (code 2000000) and the handle keeps its probed XCD placement -- a 64-sequence forward afterwards (a side-by-side
    schedule that needs the tables) is as fast as before."""
    import warnings
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    B, T = 64, 60
    imu = cu(torch_mod, synthetic.make_imu(B, T, seed=3))

    def ms(net, reps=15):
        # 40 forwards of warm-up (~40 ms), then the MEDIAN of three windows.  Round 6 measured what an idle stretch costs
        # (tools/debug/idle_probe.py, profiles/r06_idle_probe_S512.txt): under matrix load the shader clock restarts at
        # 2.17 GHz after as little as 16 ms without such load and needs ~20 ms of it to be back at 2.35 GHz -- the recovery in
        # the middle of this test is such a stretch (a synchronised run of thousands of per-step launches), so the comparison
        # starts behind a warm-up longer than the ramp.  Round 5 took the BEST of three windows here; the median does not
        # forgive a handle that is slow for good (tables off: 2.8 ms instead of 0.9).
        for _ in range(40):
            net.reset_all(); net.forward(imu, [T] * B)
        win = []
        for _w in range(3):
            torch_mod.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                net.reset_all(); net.forward(imu, [T] * B)
            torch_mod.cuda.synchronize()
            win.append(1e3 * (time.perf_counter() - t0) / reps)
        return sorted(win)[1]

    with MobilePoserNet.from_numpy(weights, smpl) as net:
        net.set_lstm_mode(1)
        before = ms(net)
        rng = np.random.Generator(np.random.PCG64(9))
        x = cu(torch_mod, (rng.standard_normal((B, 4, 132)) * 0.5).astype(np.float32))
        h0 = (rng.standard_normal((2, B, 256)) * 0.3).astype(np.float32)
        h0[1, 3, 9] = 2.5
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            net.rnn_forward("velocity", x, [4] * B, (cu(torch_mod, h0), cu(torch_mod, np.zeros_like(h0))))
        assert net.recovery_count == 1 and any("2000000" in str(i.message) for i in w)
        assert net.device_info()["placement_tables"]
        after = ms(net)
        print("64 x 60 forward: %.3f ms before, %.3f ms after a state-code recovery" % (before, after))
        if not after < 1.10 * before + 0.02:        # (seen now and then inside the whole suite, never alone: say what ran)
            names = {0: "gemm", 1: "bi256", 4: "bi512", 5: "uni", 6: "foot", 7: "per-step", 2: "ik", 3: "whole"}
            rec0 = net.recovery_count
            net.timing_enable(True)
            net.reset_all(); net.forward(imu, [T] * B); torch_mod.cuda.synchronize()
            cls = {names[c]: (net.timing_read(c)[0], round(net.timing_read(c)[1], 3)) for c in names}
            net.timing_enable(False)
            again = ms(net)
            raise AssertionError("slow after the recovery: before %.3f ms, after %.3f ms, once more %.3f ms; recoveries %d -> %d; "
                                 "device_info %s; classes of one forward (launches, ms): %s; MP_VARIANT=%r MP_WAIT_MS=%r MP_GRAPH=%r"
                                 % (before, after, again, rec0, net.recovery_count, net.device_info(), cls,
                                    os.environ.get("MP_VARIANT"), os.environ.get("MP_WAIT_MS"), os.environ.get("MP_GRAPH")))


def test_create_on_a_device_index_the_process_cannot_see(torch_mod, weights, smpl):
    from mobileposer_amd import _lib
    from mobileposer_amd.model_utils import state_dict_to_blob
    lib = _lib.load()
    blob = np.ascontiguousarray(state_dict_to_blob(weights), dtype=np.float32)
    parent = (C.c_int32 * 24)(*([-1] + [int(p) for p in smpl["kintree_table"][0][1:]]))
    J = np.ascontiguousarray(smpl["J"], dtype=np.float32).reshape(-1)
    n_dev = torch_mod.cuda.device_count()
    for bad in (n_dev, n_dev + 7, -1):
        h = C.c_void_p()
        rc = lib.mp_create(C.byref(h), bad, blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size, parent,
                           J.ctypes.data_as(C.POINTER(C.c_float)))
        assert rc == _lib.MP_ERR_INVALID and not h.value, (bad, rc)
        assert "device index" in _lib.last_error(None)
    assert torch_mod.cuda.current_device() == 0


def test_remapped_device_index_zero_in_a_subprocess():
    """HIP_VISIBLE_DEVICES renumbers the devices a process sees from 0 (what a launcher that pins one GPU per rank does): a
    handle on index 0 of such a process works, reports device 0 and the probed XCD order, and its forward matches the oracle."""
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
from oracle import mp_oracle as O
sd, smpl = synthetic.make_weights(0), synthetic.synthetic_smpl()
imu = synthetic.make_imu(20, 12, seed=2)
with MobilePoserNet.from_numpy(sd, smpl, device="cuda:0") as net:
    info = net.device_info()
    pose, joints, vel, contact = net.forward(torch.from_numpy(imu).cuda(), [12] * 20)
    rp, rj, rv, rc = O.OracleNet(sd, smpl["J"]).forward(imu, [12] * 20)
    err = float(np.abs(joints.cpu().numpy() - rj).max())
print("INFO", info["device"], info["n_cu"], int(info["xcd_round_robin"]), info["build_id"], "%%.2e" %% err, torch.cuda.device_count())
""" % REPO
    env = dict(os.environ, HIP_VISIBLE_DEVICES="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("INFO")][0].split()
    from mobileposer_amd import _lib
    assert line[1] == "0" and int(line[2]) >= 32 and line[4] == _lib.source_md5() and float(line[5]) < 1e-4 and line[6] == "1"


def test_build_id_is_the_md5_of_the_sources(net):
    from mobileposer_amd import _lib
    info = net.device_info()
    assert info["build_id"] == _lib.source_md5() == _lib.file_build_id()
    assert info["device"] == 0 and info["n_cu"] == 256


@pytest.mark.parametrize("B,T", [(256, 40), (129, 3)])
def test_fused_ik_fk_kernel_equals_the_two_kernels_bitwise(torch_mod, weights, smpl, B, T):
    """mp_forward_offline with the FK outputs requested runs r6d -> local pose -> forward kinematics as ONE kernel (mp_r6d_ik_fk,
    round 5).  Same arithmetic in the same order as mp_r6d_ik_lds + mp_fk: pose, R_global and joint positions are bit for bit
    what the stand-alone entry points give on the same r6d (frame counts that are and are not multiples of the 8 frames per block)."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    x = cu(torch_mod, synthetic.make_imu(B, T, seed=B + T))
    lens = (C.c_int32 * B)(*([T] * B))
    with MobilePoserNet.from_numpy(weights, smpl) as net:
        net.set_lstm_mode(1)
        o = _offline_buffers(torch_mod, B, T)
        net.reset_all()
        net.forward_offline_into(x, lens, o["pose"], o["joints"], o["vel"], o["contact"], o["tran"], o["rglob"], o["jglob"])
        net.reset_all()
        pose2, _j, _v, _c, r6d = net.forward(x, [T] * B, return_r6d=True)
        assert torch_mod.equal(o["pose"], pose2)
        assert torch_mod.equal(o["pose"], net._reduced_global_to_full(r6d))
        Rg, jg = net.forward_kinematics(o["pose"])
        assert torch_mod.equal(o["rglob"], Rg.reshape(o["rglob"].shape)) and torch_mod.equal(o["jglob"], jg.reshape(o["jglob"].shape))
        assert net.device_error() == 0


# ---- mp_stream_replay: N forward_online calls as one library call (evaluate.py:62-64) ---------------------------------------
def _check_online(torch_mod, got, g, keys, lo, hi, tol=TOL):
    pose, joints, root, contact = (npy(t) for t in got)
    for k in range(lo, hi):
        i = k - lo
        assert geodesic(pose[i].reshape(24, 3, 3), g[keys["pose"]][k].reshape(24, 3, 3)).max() < tol, k
        assert np.abs(joints[i][40] - g[keys["joints40"]][k]).max() < tol, k
        assert np.abs(contact[i] - g[keys["contact"]][k]).max() < tol, k
        assert np.abs(root[i] - g[keys["tran"]][k]).max() < TOL_TRAN, k


G5_KEYS = {"pose": "pose", "joints40": "joints40", "contact": "contact", "tran": "tran"}
G14_KEYS = {"pose": "on_pose", "joints40": "on_joints40", "contact": "on_contact", "tran": "on_tran"}


@pytest.mark.parametrize("split", [None, 25, 1])
def test_replay_of_the_online_goldens(torch_mod, weights, smpl, split):
    """Golden G5 (60 forward_online calls recorded from the reference) through mp_stream_replay: all 60 frames in one call, in
    two calls (state carried between them: window, velocity LSTM state, foot positions, root), and one replayed frame
    followed by 59 single ticks -- the same outputs and the same final velocity state within 1e-4 / 1 mm."""
    from conftest import load_golden
    from mobileposer_amd.net import MobilePoserNet
    g = load_golden("g5_online.npz")
    frames = cu(torch_mod, g["imu"])
    n = len(g["imu"])
    with MobilePoserNet.from_numpy(weights, smpl) as net:
        net.set_lstm_mode(1)
        net.reset()
        if split is None:
            _check_online(torch_mod, net.forward_online_replay(frames), g, G5_KEYS, 0, n)
        elif split == 1:
            _check_online(torch_mod, net.forward_online_replay(frames[:1]), g, G5_KEYS, 0, 1)
            for k in range(1, n):
                pose, joints, tran, contact = net.forward_online(frames[k])
                _check_online(torch_mod, (pose[None], joints[None], tran[None], contact[None]), g, G5_KEYS, k, k + 1)
        else:
            _check_online(torch_mod, net.forward_online_replay(frames[:split]), g, G5_KEYS, 0, split)
            st = net.stream_state(0)
            assert np.array_equal(npy(st["imu"]), g["imu"][split - 45:split] if split >= 45 else
                                  np.concatenate([np.repeat(g["imu"][:1], 45 - split, 0), g["imu"][:split]]))
            _check_online(torch_mod, net.forward_online_replay(frames[split:]), g, G5_KEYS, split, n)
        h, c = net.velocity.rnn_state
        assert np.abs(npy(h) - g["vel_h"]).max() < TOL and np.abs(npy(c) - g["vel_c"]).max() < TOL
        assert net.device_error() == 0 and net.recovery_count == 0


def test_replay_on_the_per_step_kernels_and_after_a_starved_launch(torch_mod, weights, smpl, monkeypatch):
    """mp_stream_replay's other code path: LSTM mode 0 (per-step kernels: the state of the 2 700-step velocity chain is staged
    through the plan's buffers) reproduces golden G5, and a replay whose first fused launch loses a workgroup repairs itself on
    that path (recovery on) -- same outputs, one recovery, the carried state as if nothing had happened."""
    import warnings
    from conftest import load_golden
    from mobileposer_amd.net import MobilePoserNet
    g = load_golden("g5_online.npz")
    frames = cu(torch_mod, g["imu"])
    n = len(g["imu"])
    monkeypatch.setenv("MP_WAIT_MS", "15")
    with MobilePoserNet.from_numpy(weights, smpl) as net:
        net.set_lstm_mode(0)
        net.reset()
        _check_online(torch_mod, net.forward_online_replay(frames), g, G5_KEYS, 0, n)
        h, c = net.velocity.rnn_state
        assert np.abs(npy(h) - g["vel_h"]).max() < TOL and np.abs(npy(c) - g["vel_c"]).max() < TOL
    with MobilePoserNet.from_numpy(weights, smpl) as net:
        net.set_lstm_mode(1)
        net.reset()
        _check_online(torch_mod, net.forward_online_replay(frames[:20]), g, G5_KEYS, 0, 20)
        assert net._lib.mp_debug_drop_workgroup(net._h, 8, 0, 1) == 0           # the next fused launch loses block 8
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = net.forward_online_replay(frames[20:])
        assert net.recovery_count == 1 and any("starved" in str(i.message) for i in w), [str(i.message) for i in w]
        _check_online(torch_mod, got, g, G5_KEYS, 20, n)
        h, c = net.velocity.rnn_state
        assert np.abs(npy(h) - g["vel_h"]).max() < TOL and np.abs(npy(c) - g["vel_c"]).max() < TOL
        assert net.device_error() == 0


def test_replay_of_the_trained_regime_online_golden(torch_mod, weights_trained, smpl):
    from conftest import load_golden
    from mobileposer_amd.net import MobilePoserNet
    g = load_golden("g14_trained.npz")
    with MobilePoserNet.from_numpy(weights_trained, smpl) as net:
        net.set_lstm_mode(1)
        net.reset_all()
        _check_online(torch_mod, net.forward_online_replay(cu(torch_mod, g["on_imu"])), g, G14_KEYS, 0, len(g["on_imu"]))
        h, c = net.velocity.rnn_state
        assert np.abs(npy(c) - g["on_vel_c"]).max() < TOL * max(1.0, float(np.abs(g["on_vel_c"]).max()))
        assert net.device_error() == 0


def test_replay_equals_the_ticks_and_is_faster(torch_mod, weights, smpl):
    """300 frames: the replay against the same frames fed one forward_online call at a time (fp32-noise-level differences: the
    batch shape picks other kernels), and the wall time of both."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    n = 300
    frames = cu(torch_mod, synthetic.make_imu(1, n, seed=44)[0])
    with MobilePoserNet.from_numpy(weights, smpl) as net:
        net.set_lstm_mode(1)
        net.reset_all()
        torch_mod.cuda.synchronize(); t0 = time.perf_counter()
        ticks = [net.forward_online(f) for f in frames]
        torch_mod.cuda.synchronize(); t_ticks = time.perf_counter() - t0
        ticks = [torch_mod.stack([o[i] for o in ticks]) for i in range(4)]
        h_t, c_t = net.velocity.rnn_state
        net.reset_all()
        net.last_lfoot_pos, net.last_rfoot_pos = net.feet_pos[0], net.feet_pos[1]   # (reset() keeps them, net.py:84-88)
        net.forward_online_replay(frames[:2]); net.reset_all()
        net.last_lfoot_pos, net.last_rfoot_pos = net.feet_pos[0], net.feet_pos[1]
        torch_mod.cuda.synchronize(); t0 = time.perf_counter()
        rep = net.forward_online_replay(frames)
        torch_mod.cuda.synchronize(); t_rep = time.perf_counter() - t0
        h_r, c_r = net.velocity.rnn_state
        for name, a, b, tol in (("pose", ticks[0], rep[0], 2e-5), ("joints", ticks[1], rep[1], 2e-5), ("root", ticks[2], rep[2], 2e-4),
                                ("contact", ticks[3], rep[3], 2e-5)):
            assert float((a - b).abs().max()) < tol, (name, float((a - b).abs().max()))
        assert float((h_t - h_r).abs().max()) < 2e-5 and float((c_t - c_r).abs().max()) < 2e-5
        print("300 frames: %d ticks %.1f ms (%.3f ms per tick), replay %.1f ms" % (n, 1e3 * t_ticks, 1e3 * t_ticks / n, 1e3 * t_rep))
        assert t_rep < 0.5 * t_ticks
        assert net.device_error() == 0 and net.recovery_count == 0


def test_single_sequence_kernel_and_its_velocity_wavefront(torch_mod, weights, smpl, monkeypatch):
    """B <= 16, exact-fp32: the velocity block runs both layers as ONE launch -- layer 1 reads layer 0's output behind per-wave
    progress words, one XCD per layer (mp_lstm_u8<256,*,true>; B = 1: mp_lstm_v1<256,*,true>) -- the same arithmetic as two
    launches (MP_VARIANT wf=0), bit for bit: one sequence, partly filled and full slabs, ragged lengths, carried velocity state,
    a long sequence, the zeroed-area variant, the any-placement transport; then the streaming tick and the replay chain (which
    carry the state of both layers in place).  And B = 1 on the matrix-vector kernel (mp_lstm_v1: vector-ALU dot products, DPP /
    permlane reductions, granule hand-off) against the 32-slice MFMA kernel (MP_VARIANT vec=0): another order of summation,
    equal to fp32 rounding."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    rng = np.random.default_rng(71)
    shapes = ((1, 45), (1, 1), (1, 2), (5, 31), (16, 45), (3, 900), (1, 700))
    lens = {sh: [int(v) for v in rng.integers(1, sh[1] + 1, size=sh[0])] for sh in shapes}
    frames = cu(torch_mod, synthetic.make_imu(1, 60, seed=5)[0])
    outs = {}
    for variant, remote in (("wf=0", 0), ("", 0), ("epoch_tags=0", 0), ("", 1), ("vec=0,wf=0", 0), ("vec=0", 0), ("vec=0", 1)):
        monkeypatch.setenv("MP_VARIANT", variant)
        with MobilePoserNet.from_numpy(weights, smpl) as n:
            n.set_lstm_mode(1)
            if remote:
                n.set_transport(True)
            o = []
            for B, T in shapes:
                L = list(lens[(B, T)])
                L[0] = T
                x = cu(torch_mod, synthetic.make_imu(B, T, seed=B + T))
                o += [t.clone() for t in n.forward_offline(x, L)]
                o += [t.clone() for t in n.forward_offline(x, L)]      # carried velocity state
                n.reset_all()
            for f in frames[:50]:
                o += [t.clone() for t in n.forward_online(f)]
            o += [t.clone() for t in n.velocity.rnn_state]
            n.reset_all()
            o += [t.clone() for t in n.forward_online_replay(frames)]
            o += [t.clone() for t in n.velocity.rnn_state]
            assert n.device_error() == 0 and n.recovery_count == 0
        outs[(variant, remote)] = o
    for ref, keys in ((("wf=0", 0), (("", 0), ("epoch_tags=0", 0), ("", 1))), (("vec=0,wf=0", 0), (("vec=0", 0), ("vec=0", 1)))):
        for key in keys:
            assert len(outs[ref]) == len(outs[key])
            for i, (a, b) in enumerate(zip(outs[ref], outs[key])):
                assert torch_mod.equal(a, b), (key, i, float((a - b).abs().max()))
    worst = 0.0
    for i, (a, b) in enumerate(zip(outs[("", 0)], outs[("vec=0", 0)])):
        worst = max(worst, float((a - b).abs().max()))
        assert float((a - b).abs().max()) < 2e-5, (i, float((a - b).abs().max()))
    print("mp_lstm_v1 vs mp_lstm_u8 at B = 1: max abs difference %.2e" % worst)


@pytest.mark.parametrize("B", [1, 7, 40, 200])
def test_one_frame_calls_on_a_carried_state(torch_mod, weights, smpl, B):
    """forward_offline on single frames (T = 1), five calls in a row: the velocity block starts every call from the state the
    call before left IN PLACE.  In a one-step launch of the persistent kernels nobody waits for anybody while every workgroup
    reads the whole initial h, so the step-0 operand comes from a copy (LstmDir::hin, round 5; a late workgroup could otherwise
    have found a neighbour's final state there).  Against the per-step kernels (mode 0), which ping-pong their state."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    xs = [cu(torch_mod, synthetic.make_imu(B, 1, seed=100 + i)) for i in range(5)]
    outs = {}
    with MobilePoserNet.from_numpy(weights, smpl) as n:
        for mode in (0, 1):
            n.set_lstm_mode(mode)
            n.reset_all()
            o = []
            for x in xs:
                o += [t.clone() for t in n.forward_offline(x, [1] * B)]
            o += [t.clone() for t in n.velocity.rnn_state]
            outs[mode] = o
            assert n.device_error() == 0
    assert len(outs[0]) == len(outs[1])
    for i, (a, b) in enumerate(zip(outs[0], outs[1])):
        assert float((a - b).abs().max()) < 2e-5, (i, float((a - b).abs().max()))


@pytest.fixture(scope="module")
def weights_trained():
    from mobileposer_amd.synthetic import make_weights
    return make_weights(0, profile="trained")

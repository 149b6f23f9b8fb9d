"""GPU tests at the sizes BASELINE.json's configs name, and of the facade surface the reference's callers use.

  configs[0]  evaluate.py's real call: one sequence [1, 3000, 60] through forward_offline          (a)
  configs[3]  B = 1024 x 125 (four 256-sequence launch groups per layer) vs the oracle on 32 rows    (b)
  configs[4]  S = 512 concurrent streams, 10 ticks, 4 streams followed by the oracle; masked reset   (d)
  8(e)        model built from a weight blob that already lives in HBM (the RCCL-broadcast path)     (e)
  a14         load_model on a torch.save'd state dict and on a Lightning-style checkpoint            (f)
Tolerances as everywhere: 1e-4 on network outputs / joint angles, 1 mm on root translation.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, cu, geodesic, npy

pytestmark = pytest.mark.gpu

TOL = 1e-4
TOL_TRAN = 1e-3


def test_config0_one_long_sequence_offline(torch_mod, net, weights, smpl):
    """evaluate.py:58 -- model.forward_offline(x.unsqueeze(0), [T]) with T = 3000: one slab with
    one valid row, 3000 dependent steps per layer; carried velocity state of a second call (Q1)."""
    from mobileposer_amd import synthetic
    from oracle import mp_oracle as O
    T = 3000
    imu = synthetic.make_imu(1, T, seed=61)
    ref = O.OracleNet(weights, smpl["J"])
    net.reset_all()
    for rep in range(2):                              # second pass starts from the first pass's velocity state
        net.reset()
        pose, joints, tran, contact = net.forward_offline(cu(torch_mod, imu), [T])
        ref.reset()
        rpose, rjoints, rtran, rcontact = ref.forward_offline(imu, [T])
        assert tuple(pose.shape) == (T, 24, 3, 3) and tuple(tran.shape) == (T, 3) and tuple(contact.shape) == (T, 2)
        assert np.abs(npy(joints) - rjoints).max() < TOL, rep
        assert np.abs(npy(contact) - rcontact).max() < TOL, rep
        assert geodesic(npy(pose), rpose).max() < TOL, rep
        assert np.abs(npy(tran) - rtran).max() < TOL_TRAN, rep
    assert net.device_error() == 0


def test_config3_batch_1024_vs_oracle_rows(torch_mod, net, weights, smpl):
    """BASELINE configs[3] on one GPU: 1024 sequences x 125 frames = four launch groups of 256 per layer.  The oracle runs
    on 32 rows spread over all four groups (sequences are independent), ragged lengths included."""
    from mobileposer_amd import synthetic
    from oracle import mp_oracle as O
    B, T = 1024, 125
    imu = synthetic.make_imu(B, T, seed=71)
    lengths = [T] * B
    rows = sorted(set([0, 15, 16, 255, 256, 257, 511, 512, 767, 768, 1000, 1023] + list(range(33, 1024, 50))))[:32]
    for k, r in enumerate(rows[::3]):
        lengths[r] = 1 + (37 * (k + 1)) % T
    lengths[1023] = T
    net.reset_all()
    pose, joints, tran, contact = net.forward_offline(cu(torch_mod, imu), lengths)
    assert net.device_error() == 0
    pose_h = npy(pose).reshape(B, T, 24, 3, 3)
    joints_h, tran_h, contact_h = npy(joints), npy(tran), npy(contact)
    for r in rows:
        L = lengths[r]
        ref = O.OracleNet(weights, smpl["J"])
        rp, rj, rt, rc = ref.forward_offline(imu[r:r + 1, :L], [L])
        assert np.abs(joints_h[r, :L] - rj[0]).max() < TOL, r
        assert np.abs(contact_h[r, :L] - rc).max() < TOL, r
        assert geodesic(pose_h[r, :L], rp).max() < TOL, r
        assert np.abs(tran_h[r, :L] - rt).max() < TOL_TRAN, r


def test_config3_share_of_one_gpu_in_eight_vs_oracle(torch_mod, net, weights, smpl):
    """BASELINE configs[3] split over 8 GPUs (dist.shard_range): each rank gets 128 x 125 -- on the exact-fp32 path the
    half-chip schedule (pose layer 1 beside the velocity layers with the foot-contact rider).  The whole shard, ragged,
    forward_offline (network + FK-free translation solver), every row against the oracle."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.dist import shard_range
    from oracle import mp_oracle as O
    lo, hi = shard_range(1024, 5, 8)                                   # rank 5's rows of the global batch
    B, T = hi - lo, 125
    assert B == 128
    imu = synthetic.make_imu(1024, T, seed=71)[lo:hi]
    lengths = [T - (11 * b) % 97 for b in range(B)]
    lengths[7] = T
    net.reset_all()
    pose, joints, tran, contact = net.forward_offline(cu(torch_mod, imu), lengths)
    assert net.device_error() == 0 and net.recovery_count == 0
    pose_h = npy(pose).reshape(B, T, 24, 3, 3)
    joints_h, tran_h, contact_h = npy(joints), npy(tran), npy(contact)
    ref = O.OracleNet(weights, smpl["J"])
    rpose, rjoints, rvel, rcontact = ref.forward(imu, lengths)       # the batched oracle forward; solver per row below
    rpose = np.asarray(rpose).reshape(B, T, 24, 3, 3)
    for r in range(B):
        L = lengths[r]
        assert np.abs(joints_h[r, :L] - rjoints[r, :L]).max() < TOL, r
        assert np.abs(contact_h[r, :L] - rcontact[r, :L]).max() < TOL, r
        assert geodesic(pose_h[r, :L], rpose[r, :L]).max() < TOL, r
        rt = O.translate_offline(rjoints[r, :L].reshape(L, 24, 3), rvel[r, :L], rcontact[r, :L], ref.floor_y)
        assert np.abs(tran_h[r, :L] - rt).max() < TOL_TRAN, r


def test_config4_512_streams_and_masked_reset(torch_mod, net, weights, smpl):
    """BASELINE configs[4] on one GPU: 512 concurrent streams ticked 10 times; streams 0, 17, 255 and 511 are followed
    by the oracle's forward_online.  Then reset() for a subset (mask): those streams restart like fresh ones (window
    refilled, root back at 0; the velocity LSTM state is kept, net.py:84-88) while the others continue."""
    from mobileposer_amd import synthetic
    from oracle import mp_oracle as O
    S, n1, n2 = 512, 10, 3
    watch = [0, 17, 255, 511]
    frames = synthetic.make_imu(S, n1 + n2, seed=83)
    refs = {s: O.OracleNet(weights, smpl["J"]) for s in watch}
    net.reset_all()
    net.stream_create(S)

    def tick(k):
        pose, joints, root, contact = net.stream_step(cu(torch_mod, frames[:, k]))
        for s in watch:
            rp, rj, rr, rc = refs[s].forward_online(frames[s, k])
            assert geodesic(npy(pose[s]).reshape(24, 3, 3), rp.reshape(24, 3, 3)).max() < TOL, (s, k)
            assert np.abs(npy(joints[s]) - rj).max() < TOL, (s, k)
            assert np.abs(npy(contact[s]) - rc).max() < TOL, (s, k)
            assert np.abs(npy(root[s]) - rr).max() < TOL_TRAN, (s, k)

    for k in range(n1):
        tick(k)
    # the reference's state attributes, read back from the device (net.py:59-64,205-208)
    st = net.stream_state(17)
    assert np.abs(npy(st["imu"]) - refs[17].imu).max() == 0
    assert abs(st["current_root_y"] - refs[17].current_root_y) < TOL_TRAN
    assert np.abs(npy(st["last_root_pos"]) - refs[17].last_root_pos).max() < TOL_TRAN
    assert np.abs(npy(st["last_lfoot_pos"]) - refs[17].last_lfoot_pos).max() < TOL
    # reset() for streams 17 and 511 only
    mask = np.zeros(S, dtype=bool)
    mask[[17, 511]] = True
    net.stream_reset(mask)
    for s in (17, 511):
        refs[s].reset()
    assert net.stream_state(17)["imu"] is None and net.stream_state(17)["current_root_y"] == 0
    assert net.stream_state(0)["imu"] is not None
    for k in range(n1, n1 + n2):
        tick(k)
    assert net.device_error() == 0


def test_state_attributes_single_stream(torch_mod, net):
    """imu / current_root_y / last_root_pos of the reference object (net.py:59-64) on the single-stream facade."""
    from mobileposer_amd import synthetic
    frames = synthetic.make_imu(1, 3, seed=5)[0]
    net.reset()
    assert net.imu is None and net.current_root_y == 0 and float(net.last_root_pos.abs().max()) == 0
    for f in frames:
        _, _, root, _ = net.forward_online(cu(torch_mod, f))
    assert tuple(net.imu.shape) == (45, 60)
    assert np.abs(npy(net.imu[-1]) - frames[-1]).max() == 0 and np.abs(npy(net.imu[0]) - frames[0]).max() == 0
    assert torch_mod.equal(net.last_root_pos, root)
    assert abs(net.current_root_y - float(root[1])) < 1e-5
    net.reset()
    assert net.imu is None and net.current_root_y == 0


def test_from_device_blob_matches_host_blob(torch_mod, weights, smpl):
    """rank r builds its model from the broadcast blob in HBM (mp_create_from_device)."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.model_utils import state_dict_to_blob
    from mobileposer_amd.net import MobilePoserNet
    x = cu(torch_mod, synthetic.make_imu(5, 30, seed=2))
    blob = torch_mod.from_numpy(state_dict_to_blob(weights)).cuda()
    with MobilePoserNet.from_numpy(weights, smpl) as a, MobilePoserNet.from_device_blob(blob, smpl) as b:
        oa, ob = a.forward(x, [30] * 5), b.forward(x, [30] * 5)
        for u, v in zip(oa, ob):
            assert torch_mod.equal(u, v)
        assert set(b.state_dict()) == set(weights)
        assert np.array_equal(b.state_dict()["joints.joints.linear2.bias"].numpy(), weights["joints.joints.linear2.bias"])


def test_load_model_state_dict_and_checkpoint(torch_mod, weights, smpl, tmp_path):
    """utils/model_utils.py:6-15: a torch.save'd state dict (combine_weights.py:53-56), and the Lightning-checkpoint
    fallback (weights under 'state_dict', here also behind a wrapper prefix and next to unrelated entries)."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.model_utils import load_model
    from mobileposer_amd.net import MobilePoserNet
    sd = {k: torch_mod.from_numpy(v) for k, v in weights.items()}
    p1, p2, p3 = tmp_path / "weights.pth", tmp_path / "lightning.ckpt", tmp_path / "wrapped.ckpt"
    torch_mod.save(sd, p1)
    torch_mod.save({"epoch": 3, "state_dict": dict(sd), "hyper_parameters": {"finetune": False}}, p2)
    torch_mod.save({"state_dict": {**{"model." + k: v for k, v in sd.items()}, "loss.weight": torch_mod.ones(1)}}, p3)
    x = cu(torch_mod, synthetic.make_imu(2, 20, seed=9))
    with MobilePoserNet.from_numpy(weights, smpl) as ref:
        want = ref.forward(x, [20, 20])
    for p in (p1, p2, p3):
        m = load_model(str(p), smpl=smpl)
        try:
            got = m.forward(x, [20, 20])
            for u, v in zip(want, got):
                assert torch_mod.equal(u, v), p.name
        finally:
            m.close()
    broken = dict(sd)
    del broken["velocity.vel.linear2.bias"]
    torch_mod.save(broken, p1)
    with pytest.raises(KeyError):
        load_model(str(p1), smpl=smpl)


def test_graph_replay_equals_eager_in_subprocess():
    """Opt-in hipGraph mode (mp_set_graph_mode(h, 1)): replay == eager, bitwise, for the batch path and the streaming
    tick.  In its own process with GPU_MAX_HW_QUEUES=8: the multi-branch graph executor of this ROCm's HIP runtime can
    segfault in hipGraphLaunch for unlucky hardware-queue placements (profiles/r02_hipgraph_segv.md)."""
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from mobileposer_amd import synthetic
from mobileposer_amd.net import MobilePoserNet
sd, smpl = synthetic.make_weights(0), synthetic.synthetic_smpl()
x = torch.from_numpy(synthetic.make_imu(64, 50, seed=3)).cuda()
fr = torch.from_numpy(synthetic.make_imu(4, 6, seed=4)).cuda()
for mode in (1, 3):
    outs = {}
    for graph in (0, 1):
        with MobilePoserNet.from_numpy(sd, smpl) as m:
            m.set_lstm_mode(mode); m.set_graph_mode(graph)
            o = [t.clone() for t in m.forward_offline(x, [50] * 64)]
            o += [t.clone() for t in m.forward_offline(x, [50] * 64)]      # replay, carried velocity state
            m.velocity.rnn_state = None
            m.stream_create(4)
            for k in range(6):
                o += [t.clone() for t in m.stream_step(fr[:, k].contiguous())]
            assert m.device_error() == 0
            outs[graph] = o
    assert len(outs[0]) == len(outs[1])
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b), mode
# a velocity state that has to grow in the middle (its re-allocation drops every captured graph), three passes
shapes = ((256, 40), (40, 25), (300, 9), (17, 60))
xs = {sh: torch.from_numpy(synthetic.make_imu(sh[0], sh[1], seed=sum(sh))).cuda() for sh in shapes}
outs = {}
for graph in (0, 1):
    with MobilePoserNet.from_numpy(sd, smpl) as m:
        m.set_graph_mode(graph)
        o = []
        for _ in range(3):
            for sh in shapes:
                m.reset_all()
                for _call in range(2):
                    o += [t.clone() for t in m.forward_offline(xs[sh], [sh[1]] * sh[0])]
        assert m.device_error() == 0
        outs[graph] = o
for a, b in zip(outs[0], outs[1]):
    assert torch.equal(a, b)
print("GRAPH_OK")
''' % REPO
    # (round 5: graph mode 1 means mode 2 unless MP_GRAPH_MULTIBRANCH=1 -- this test is the one that asks for the real thing)
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", MP_GRAPH_MULTIBRANCH="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "GRAPH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("mode", [1, 3])
def test_single_branch_graph_equals_eager_at_baseline_sizes(torch_mod, weights, smpl, mode):
    """Graph mode 2 (every launch captured on one stream: a single-branch graph, no parallel streams for the runtime's graph
    executor to assign -- no GPU_MAX_HW_QUEUES workaround, in-process): replay == eager bitwise at the sizes BASELINE.json
    names -- the 256 x 125 batch (forward + FK-less offline solver, carried velocity state) and the S = 512 streaming tick
    (configs[4]) -- and a caller that hands in fresh tensors every call (the facade) keeps replaying ONE graph."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    B, T, S, ticks = 256, 125, 512, 5
    x = cu(torch_mod, synthetic.make_imu(B, T, seed=3))
    fr = cu(torch_mod, synthetic.make_imu(S, ticks, seed=4))
    outs = {}
    for graph in (0, 2):
        with MobilePoserNet.from_numpy(weights, smpl) as m:
            m.set_lstm_mode(mode)
            m.set_graph_mode(graph)
            o = [t.clone() for t in m.forward_offline(x.clone(), [T] * B)]
            o += [t.clone() for t in m.forward_offline(x.clone(), [T] * B)]     # replay; velocity state carried (Q1)
            o += [t.clone() for t in m.forward_offline(x.clone(), [T] * B)]
            m.velocity.rnn_state = None
            m.stream_create(S)
            for k in range(ticks):
                o += [t.clone() for t in m.stream_step(fr[:, k].contiguous())]
            assert m.device_error() == 0
            outs[graph] = o
    assert len(outs[0]) == len(outs[2])
    for i, (a, b) in enumerate(zip(outs[0], outs[2])):
        assert torch_mod.equal(a, b), i


def test_graphs_survive_a_growing_velocity_state(torch_mod, weights, smpl, gmode=2):
    """Regression (round 3): the carried velocity state is re-allocated when a larger batch arrives; captured graphs hold the
    old buffers' addresses, and the new buffer can land on the old address -- a stale graph then matched its key again and
    wrote through a freed pointer (GPU memory fault).  All graphs go when those buffers go: shapes that force the
    re-allocation in the middle, replayed over three passes, bitwise equal to eager (single-branch graphs here; the
    multi-branch mode runs the same shapes in test_graph_replay_equals_eager_in_subprocess)."""
    from mobileposer_amd import synthetic
    from mobileposer_amd.net import MobilePoserNet
    shapes = ((256, 40), (40, 25), (300, 9), (17, 60))
    xs = {sh: cu(torch_mod, synthetic.make_imu(sh[0], sh[1], seed=sum(sh))) for sh in shapes}
    outs = {}
    for graph in (0, gmode):
        with MobilePoserNet.from_numpy(weights, smpl) as m:
            m.set_graph_mode(graph)
            o = []
            for _ in range(3):
                for sh in shapes:
                    m.reset_all()
                    for _call in range(2):
                        o += [t.clone() for t in m.forward_offline(xs[sh], [sh[1]] * sh[0])]
            assert m.device_error() == 0
            outs[graph] = o
    for a, b in zip(outs[0], outs[gmode]):
        assert torch_mod.equal(a, b)


def test_live_frame_kernel_golden_and_host_version(torch_mod, weights, smpl):
    """mp_live_form_frames (csrc/mp_live.hip) against golden G10 -- frames computed with the reference's own math functions,
    live_demo.py:213-236 -- and, with a different calibration per stream and every device combo, against the host version."""
    from conftest import load_golden
    from mobileposer_amd import live
    from mobileposer_amd.config import amass
    from mobileposer_amd.net import MobilePoserNet
    g = load_golden("g10_live.npz")
    with MobilePoserNet.from_numpy(weights, smpl) as m:
        F = g["fq"].shape[0]
        rep = lambda a: torch_mod.from_numpy(np.ascontiguousarray(np.broadcast_to(a, (F,) + a.shape)))
        got = m.live_form_frames(torch_mod.from_numpy(g["fq"]), torch_mod.from_numpy(g["fa"]), rep(g["smpl2imu"]),
                                 rep(g["device2bone"]), rep(g["acc_offsets"].reshape(5, 3)), live.combo_keep_mask("lw_rp"))
        assert np.abs(npy(got) - g["imu_input"]).max() < 1e-5
        rng = np.random.default_rng(12)
        S = 37
        cals = [live.Calibration.from_measurements(torch_mod.from_numpy(rng.standard_normal(4)).float(),
                                                   torch_mod.from_numpy(rng.standard_normal((5, 4))).float(),
                                                   torch_mod.from_numpy(rng.standard_normal((5, 3)) * 9.8).float()) for _ in range(S)]
        q = torch_mod.from_numpy(rng.standard_normal((S, 5, 4)) * 3).float()
        a = torch_mod.from_numpy(rng.standard_normal((S, 5, 3)) * 9.8).float()
        M = torch_mod.stack([c.smpl2imu for c in cals])
        D = torch_mod.stack([c.device2bone for c in cals])
        O = torch_mod.stack([c.acc_offsets.reshape(5, 3) for c in cals])
        for combo in amass.combos:
            got = npy(m.live_form_frames(q, a, M, D, O, live.combo_keep_mask(combo)))
            want = torch_mod.cat([live.form_frame(c, q[i][None], a[i][None], combo) for i, c in enumerate(cals)]).numpy()
            assert np.abs(got - want).max() < 2e-5, combo

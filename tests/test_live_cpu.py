"""Live front-end math (mobileposer_amd/live.py) against values computed with the reference's own math functions
(golden G10) and round trips of the wire formats.  CPU only."""
import numpy as np
import torch

from conftest import load_golden
from mobileposer_amd import live


def test_g10_calibration_and_frame_formation():
    g = load_golden("g10_live.npz")
    cal = live.Calibration.from_measurements(torch.from_numpy(g["ref_q"]), torch.from_numpy(g["tq"]), torch.from_numpy(g["ta"]))
    assert np.abs(cal.smpl2imu.numpy() - g["smpl2imu"]).max() < 1e-6
    assert np.abs(cal.device2bone.numpy() - g["device2bone"]).max() < 1e-6
    assert np.abs(cal.acc_offsets.numpy() - g["acc_offsets"]).max() < 1e-5
    frames = live.form_frame(cal, torch.from_numpy(g["fq"]), torch.from_numpy(g["fa"]), combo="lw_rp")
    assert tuple(frames.shape) == (7, 60)
    assert np.abs(frames.numpy() - g["imu_input"]).max() < 1e-5
    # devices outside the combo are zero (live_demo.py:229-236)
    assert float(frames[:, 3:9].abs().max()) == 0.0


def test_g10_axis_angle():
    g = load_golden("g10_live.npz")
    aa = live.rotation_matrix_to_axis_angle(torch.from_numpy(g["rot"]))
    assert np.abs(aa.numpy() - g["axis_angle"]).max() < 1e-5
    # rotation by pi about a coordinate axis and the identity
    R = torch.tensor([[[1.0, 0, 0], [0, -1, 0], [0, 0, -1]], [[1.0, 0, 0], [0, 1, 0], [0, 0, 1]]])
    out = live.rotation_matrix_to_axis_angle(R)
    assert np.allclose(out[0].abs().numpy(), [np.pi, 0, 0], atol=1e-5) and np.allclose(out[1].numpy(), 0)


def test_packet_round_trip_and_output_format():
    rng = np.random.default_rng(0)
    acc = rng.standard_normal((5, 3))
    quat_xyzw = rng.standard_normal((5, 4))
    pkt = live.encode_packet(acc, quat_xyzw)
    a, q = live.parse_packet(pkt)
    assert np.allclose(a, -9.8 * acc, rtol=1e-5) and np.allclose(q, quat_xyzw[:, [3, 0, 1, 2]], rtol=1e-5)
    out = live.format_output(torch.eye(3).repeat(24, 1, 1), torch.tensor([0.5, -1.0, 2.0])).decode()
    pose_s, tran_s = out.rstrip("$").split("#")
    assert len(pose_s.split(",")) == 72 and [float(v) for v in tran_s.split(",")] == [0.5, -1.0, 2.0]

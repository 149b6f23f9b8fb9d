"""Live streaming front-end math: what sits between the sensor packets and
``forward_online`` / ``mp_stream_step`` in the reference's live demo, as pure host-side functions.

Mirrors, formula for formula: the UDP packet format ``acc#quat$`` (live_demo.py:64-75, sender side
utils/socket_utils.py:19-34), T-pose calibration (live_demo.py:161-174), per-frame frame formation
(live_demo.py:213-236: sensor -> SMPL frame, channel re-order [1,4,3,0,2], /acc_scale, combo mask) and the
``pose#tran$`` output string (live_demo.py:243-256).  No sockets, threads or UI here -- those are out of scope;
``LiveSession`` feeds S calibrated sensor sets into the GPU streaming step.
"""
import numpy as np
import torch

from .config import amass

CHANNEL_ORDER = [1, 4, 3, 0, 2]          # live_demo.py:219-220, combiner.py:14-17


def quaternion_to_rotation_matrix(q):
    """articulate/math/angular.py:224-236: (unnormalised) wxyz quaternions [...,4] -> [N,3,3]."""
    q = torch.as_tensor(q, dtype=torch.float32).reshape(-1, 4)
    q = q / q.norm(dim=1, keepdim=True)
    a, b, c, d = q[:, 0:1], q[:, 1:2], q[:, 2:3], q[:, 3:4]
    r = torch.cat((-2 * c * c - 2 * d * d + 1, 2 * b * c - 2 * a * d, 2 * a * c + 2 * b * d,
                   2 * b * c + 2 * a * d, -2 * b * b - 2 * d * d + 1, 2 * c * d - 2 * a * b,
                   2 * b * d - 2 * a * c, 2 * a * b + 2 * c * d, -2 * b * b - 2 * c * c + 1), dim=1)
    return r.view(-1, 3, 3)


def rotation_matrix_to_axis_angle(r):
    """articulate/math/angular.py:154-164 (the reference calls cv2.Rodrigues per matrix): log map of SO(3)."""
    r = torch.as_tensor(r, dtype=torch.float64).reshape(-1, 3, 3)
    w = torch.stack((r[:, 2, 1] - r[:, 1, 2], r[:, 0, 2] - r[:, 2, 0], r[:, 1, 0] - r[:, 0, 1]), dim=1) * 0.5
    s = w.norm(dim=1)
    c = ((r[:, 0, 0] + r[:, 1, 1] + r[:, 2, 2]) - 1.0) * 0.5
    theta = torch.atan2(s, c)
    out = torch.zeros_like(w)
    small = s < 1e-8
    reg = ~small
    out[reg] = w[reg] * (theta[reg] / s[reg]).unsqueeze(1)
    # theta ~ pi: axis from the symmetric part (w vanishes there)
    near_pi = small & (c < 0)
    if near_pi.any():
        rr = r[near_pi]
        d = torch.stack((rr[:, 0, 0], rr[:, 1, 1], rr[:, 2, 2]), dim=1)
        ax = torch.sqrt(torch.clamp((d + 1.0) * 0.5, min=0.0))
        k = ax.argmax(dim=1)
        sign = torch.ones_like(ax)
        for i in range(rr.shape[0]):
            kk = int(k[i])
            for jj in range(3):
                if jj != kk and rr[i, kk, jj] + rr[i, jj, kk] < 0:
                    sign[i, jj] = -1.0
        out[near_pi] = ax * sign * np.pi
    return out.float()


def parse_packet(data):
    """live_demo.py:64-75: ``'ax,ay,az,...#qw,qx,qy,qz,...$'`` -> (acc [n,3] in m/s^2 = -9.8 * value, quat [n,4] wxyz)."""
    s = data.decode("utf-8") if isinstance(data, (bytes, bytearray)) else data
    a = np.array(s.split("#")[0].split(",")).astype(np.float64)
    q = np.array(s.split("#")[1].strip("$").split(",")).astype(np.float64)
    return -9.8 * a.reshape(-1, 3), q.reshape(-1, 4)


def encode_packet(acc, quat_xyzw):
    """utils/socket_utils.py:19-34 (sender): 5 sensors, quaternions re-ordered xyzw -> wxyz."""
    a = np.asarray(acc)[:5]
    o = np.asarray(quat_xyzw)[:5][:, [3, 0, 1, 2]]
    return (','.join('%g' % v for v in a.flatten()) + '#' + ','.join('%g' % v for v in o.flatten()) + '$').encode("utf8")


def format_output(pose, tran):
    """live_demo.py:243-256: 72 axis-angle numbers '#' 3 translation numbers '$'."""
    aa = rotation_matrix_to_axis_angle(torch.as_tensor(pose).reshape(-1, 3, 3).cpu()).reshape(72)
    t = torch.as_tensor(tran).reshape(3).cpu()
    return (','.join('%g' % v for v in aa) + '#' + ','.join('%g' % v for v in t) + '$').encode('utf8')


class Calibration:
    """live_demo.py:161-174."""

    def __init__(self, smpl2imu, device2bone, acc_offsets):
        self.smpl2imu, self.device2bone, self.acc_offsets = smpl2imu, device2bone, acc_offsets

    @classmethod
    def from_measurements(cls, ref_quat, tpose_quats, tpose_accs):
        """ref_quat [4]: sensor 1 aligned with the body frame; tpose_quats [n,4], tpose_accs [n,3]: T-pose means."""
        smpl2imu = quaternion_to_rotation_matrix(ref_quat).view(3, 3).t()                  # :164
        oris = quaternion_to_rotation_matrix(tpose_quats)                                   # :172
        device2bone = smpl2imu.matmul(oris).transpose(1, 2).matmul(torch.eye(3))            # :173
        acc_offsets = smpl2imu.matmul(torch.as_tensor(tpose_accs, dtype=torch.float32).unsqueeze(-1))   # :174
        return cls(smpl2imu, device2bone, acc_offsets)


def form_frame(cal, quat_raw, acc_raw, combo='lw_rp', n_imus=5):
    """live_demo.py:213-236: raw wxyz quaternions [F,5,4] / accelerations [F,5,3] -> network input [F,60]."""
    ori_raw = quaternion_to_rotation_matrix(quat_raw).view(-1, n_imus, 3, 3)
    acc_raw = torch.as_tensor(acc_raw, dtype=torch.float32)
    glb_acc = (cal.smpl2imu.matmul(acc_raw.view(-1, n_imus, 3, 1)) - cal.acc_offsets).view(-1, n_imus, 3)
    glb_ori = cal.smpl2imu.matmul(ori_raw).matmul(cal.device2bone)
    _acc = glb_acc.view(-1, 5, 3)[:, CHANNEL_ORDER] / amass.acc_scale
    _ori = glb_ori.view(-1, 5, 3, 3)[:, CHANNEL_ORDER]
    acc = torch.zeros_like(_acc)
    ori = torch.zeros_like(_ori)
    c = amass.combos[combo]
    acc[:, c] = _acc[:, c]
    ori[:, c] = _ori[:, c]
    return torch.cat([acc.flatten(1), ori.flatten(1)], dim=1)


class LiveSession:
    """S calibrated sensor sets -> one GPU streaming tick (mp_stream_step) per frame set."""

    def __init__(self, model, calibrations, combo='lw_rp'):
        self.model, self.cals, self.combo = model, list(calibrations), combo
        model.stream_create(len(self.cals))

    def tick(self, quats, accs):
        """quats [S,5,4], accs [S,5,3] (one raw sample per stream) -> (pose [S,24,9], root_pos [S,3], packets)."""
        frames = torch.cat([form_frame(c, torch.as_tensor(quats[i])[None], torch.as_tensor(accs[i])[None], self.combo)
                            for i, c in enumerate(self.cals)])
        pose, _joints, root, _contact = self.model.stream_step(frames)
        packets = [format_output(pose[i], root[i]) for i in range(len(self.cals))]
        return pose, root, packets

"""Live streaming front-end: what sits between the sensor packets and
``forward_online`` / ``mp_stream_step`` in the reference's live demo.  The per-tick arithmetic for S streams is a HIP
kernel (csrc/mp_live.hip, ``LiveSession``); calibration, wire formats and a single-set host version are here.

Contracts followed: the UDP packet format ``acc#quat$`` (live_demo.py:64-75, sender side
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
    """(Unnormalised) wxyz quaternions [...,4] -> rotation matrices [N,3,3] (contract of articulate/math/angular.py:224-236).
    With unit q = (w, v):  R = (w^2 - |v|^2) I + 2 v v^T + 2 w [v]x."""
    q = torch.as_tensor(q, dtype=torch.float32).reshape(-1, 4)
    q = q / q.norm(dim=1, keepdim=True)
    w, v = q[:, 0], q[:, 1:]
    eye = torch.eye(3, dtype=q.dtype).expand(q.shape[0], 3, 3)
    cross = torch.zeros(q.shape[0], 3, 3, dtype=q.dtype)                      # [v]x
    cross[:, 0, 1], cross[:, 0, 2], cross[:, 1, 2] = -v[:, 2], v[:, 1], -v[:, 0]
    cross = cross - cross.transpose(1, 2)
    return (w * w - (v * v).sum(dim=1)).view(-1, 1, 1) * eye + 2.0 * v.unsqueeze(2) * v.unsqueeze(1) \
        + 2.0 * w.view(-1, 1, 1) * cross


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


def combo_keep_mask(combo):
    """Bit k set = network slot k (config.py:60-73) is part of the device combo."""
    m = 0
    for k in amass.combos[combo]:
        m |= 1 << k
    return m


def form_frame(cal, quat_raw, acc_raw, combo='lw_rp', n_imus=5):
    """Host version of the frame formation (contract of live_demo.py:213-236): raw wxyz quaternions [F,5,4] and
    accelerations [F,5,3] of ONE calibrated sensor set -> network input [F,60].  Per sensor: orientation
    smpl2imu . R(q) . device2bone, acceleration smpl2imu . a - offset; then the sensors are gathered into network slot order
    (CHANNEL_ORDER), accelerations divided by acc_scale, and slots outside the combo zeroed.
    (LiveSession does the same for S sensor sets in one GPU launch, mp_live_form_frames.)"""
    F = torch.as_tensor(quat_raw).reshape(-1, n_imus, 4).shape[0]
    R = quaternion_to_rotation_matrix(quat_raw).view(F, n_imus, 3, 3)
    a = torch.as_tensor(acc_raw, dtype=torch.float32).reshape(F, n_imus, 3)
    M = cal.smpl2imu
    ori = torch.einsum('ab,fsbc,scd->fsad', M, R, cal.device2bone)
    acc = torch.einsum('ab,fsb->fsa', M, a) - cal.acc_offsets.reshape(1, n_imus, 3)
    keep = torch.zeros(n_imus)
    keep[amass.combos[combo]] = 1.0
    acc = acc[:, CHANNEL_ORDER] / amass.acc_scale * keep.view(1, n_imus, 1)
    ori = ori[:, CHANNEL_ORDER] * keep.view(1, n_imus, 1, 1)
    return torch.cat((acc.reshape(F, -1), ori.reshape(F, -1)), dim=1)


class LiveSession:
    """S calibrated sensor sets -> one GPU launch that forms the S frames (mp_live_form_frames) -> one streaming tick
    (mp_stream_step).  The calibrations live on the device for the lifetime of the session."""

    def __init__(self, model, calibrations, combo='lw_rp'):
        self.model, self.cals, self.combo = model, list(calibrations), combo
        self.keep = combo_keep_mask(combo)
        dev = model.device
        self._M = torch.stack([c.smpl2imu for c in self.cals]).float().contiguous().to(dev)                 # [S,3,3]
        self._D = torch.stack([c.device2bone for c in self.cals]).float().contiguous().to(dev)              # [S,5,3,3]
        self._O = torch.stack([c.acc_offsets.reshape(5, 3) for c in self.cals]).float().contiguous().to(dev)  # [S,5,3]
        model.stream_create(len(self.cals))

    def frames(self, quats, accs):
        """quats [S,5,4], accs [S,5,3] (one raw sample per stream) -> frames [S,60] on the device."""
        return self.model.live_form_frames(torch.as_tensor(quats), torch.as_tensor(accs), self._M, self._D, self._O, self.keep)

    def tick(self, quats, accs):
        """-> (pose [S,24,9], root_pos [S,3], packets)."""
        pose, _joints, root, _contact = self.model.stream_step(self.frames(quats, accs))
        pose_h, root_h = pose.cpu(), root.cpu()
        packets = [format_output(pose_h[i], root_h[i]) for i in range(len(self.cals))]
        return pose, root, packets

"""Dataset -> IMU-frame formation for evaluation: mirror of ``PoseDataset`` in eval mode
(data.py:45-107 of the reference) -- the step immediately before the hot path.

Host-side, load-time logic (torch CPU tensors).  On-disk format is the reference's: a ``.pt`` dict of lists
``acc [N,6,3]``, ``ori [N,6,3,3]``, ``pose [N,24,3,3]``, ``tran [N,3]`` (process.py:116-127,285-295).
Every sequence is expanded into the 12 device-location combos of ``amass.combos`` (config.py:60-73) by
zero-masking absent devices (data.py:69-76); items are ``(imu [T,60], pose_r6d [T,144], joint [T,24,3], tran [T,3])``.
"""
import torch

from .config import amass


def rotation_matrix_to_r6d(r):
    """articulate/math/angular.py:185-192: first two columns of R, column-major."""
    return r.reshape(-1, 3, 3)[:, :, :2].transpose(1, 2).clone().reshape(-1, 6)


class PoseDataset:
    def __init__(self, data, fk=None, combos=None):
        """``data``: path to a ``.pt`` file or an already-loaded dict of lists.
        ``fk``: callable pose[N,24,3,3] -> (R_global, joint[N,24,3]) giving the ground-truth joints the
        reference computes at data.py:64 (e.g. ``MobilePoserNet.forward_kinematics``); None -> joints omitted."""
        if isinstance(data, (str, bytes)) or hasattr(data, "__fspath__"):
            data = torch.load(data, map_location="cpu")
        self.combos = list((combos or amass.combos).items())
        self.items = []
        poses = data["pose"]
        for acc, ori, pose, tran in zip(data["acc"], data["ori"], poses, data["tran"]):
            acc = torch.as_tensor(acc).float()
            ori = torch.as_tensor(ori).float()
            pose = torch.as_tensor(pose).float().view(-1, 24, 3, 3)
            tran = torch.as_tensor(tran).float()
            acc, ori = acc[:, :5] / amass.acc_scale, ori[:, :5]                       # data.py:62
            joint = None
            if fk is not None:
                joint = fk(pose)[1].detach().cpu().view(-1, 24, 3)                       # data.py:63-66
            for _, c in self.combos:                                                    # data.py:70-76
                combo_acc = torch.zeros_like(acc)
                combo_ori = torch.zeros_like(ori)
                combo_acc[:, c] = acc[:, c]
                combo_ori[:, c] = ori[:, c]
                imu = torch.cat([combo_acc.flatten(1), combo_ori.flatten(1)], dim=1)    # [N,15] | [N,45] -> [N,60]
                pose_r6d = rotation_matrix_to_r6d(pose).reshape(-1, 24 * 6)             # data.py:98 (local pose kept)
                self.items.append((imu, pose_r6d, joint, tran))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, idx):
        return self.items[idx]

"""Dataset -> IMU-frame formation for evaluation: the ``PoseDataset`` the reference's ``evaluate.py`` builds
(``PoseDataset(fold='test', evaluate='dip')``, data.py:19-107) -- the step immediately before the hot path.

Host-side, load-time logic (torch CPU tensors).  On-disk format is the reference's: ``paths.processed_datasets/eval/<file>``
(config.py:33-34, datasets.test_datasets config.py:104-108), a ``.pt`` dict of lists ``acc [N,6,3]``, ``ori [N,6,3,3]``,
``pose [N,24,3,3]``, ``tran [N,3]`` (process.py:116-127,285-295).  Every sequence appears once per device-location combo of
``amass.combos`` (config.py:60-73): the accelerations / orientations of the devices a combo lacks are zero, the 60-d frame is
[5 x 3 accelerations / acc_scale | 5 x 3 x 3 orientations]; items are ``(imu [T,60], pose_r6d [T,144], joint [T,24,3],
tran [T,3])`` with the ground-truth pose kept LOCAL (data.py:65) and its joints from forward kinematics (data.py:64).

Only the evaluation fold is in scope; the training fold (windows of 125, velocity / foot-contact
targets, data.py:82-93) belongs to training and raises.
"""
import numpy as np
import torch

from .config import amass, datasets, paths
from .model_utils import UntrustedFileError, safe_torch_load


def rotation_matrix_to_r6d(r):
    """articulate/math/angular.py:185-192: the six numbers are the first two COLUMNS of R, column after column."""
    m = torch.as_tensor(r).reshape(-1, 3, 3)
    return torch.cat((m[:, :, 0], m[:, :, 1]), dim=1)


def combo_masks(combos=None):
    """[n_combos, 5] 0/1 table: which of the 5 device slots a combo keeps (config.py:60-73)."""
    combos = combos if combos is not None else amass.combos
    m = torch.zeros(len(combos), 5)
    for k, slots in enumerate(combos.values()):
        m[k, list(slots)] = 1.0
    return m


class PoseDataset:
    def __init__(self, fold='train', evaluate=None, finetune=None, data=None, fk=None, combos=None, smpl=None, trusted=False):
        """Reference signature ``PoseDataset(fold, evaluate, finetune)`` (data.py:19).  Extras, all optional: ``data`` -- an
        already-loaded dataset dict or a path instead of the configured file; ``fk`` -- callable pose -> (R_global, joint)
        for the ground-truth joints (e.g. ``MobilePoserNet.forward_kinematics``); default: what the reference does at
        data.py:24,64 -- a ``ParametricModel`` of ``smpl`` / ``paths.smpl_file`` (synthetic body when that file is absent)
        whose forward kinematics run on the GPU (mp_fk); ``combos`` -- a subset of ``amass.combos``."""
        self.fold, self.evaluate, self.finetune = fold, evaluate, finetune
        self._trusted = trusted                  # model_utils.safe_torch_load: full unpickling only for trusted files
        self.combos = list((combos if combos is not None else amass.combos).items())
        self.bodymodel = None
        self._fk = fk if fk is not None else self._body_fk(smpl)
        self.data = {k: [] for k in ('imu_inputs', 'pose_outputs', 'joint_outputs', 'tran_outputs')}
        for file_data in self._sources(data):
            self._add_file(file_data, combo_masks(dict(self.combos)))

    # ---- where the sequences come from (data.py:27-55) ------------------------------------------------------------
    def _sources(self, data):
        if data is not None:
            yield safe_torch_load(data, self._trusted) if not isinstance(data, dict) else data
            return
        if self.fold == 'test':
            if self.evaluate not in datasets.test_datasets:
                raise ValueError(f"Test dataset: {self.evaluate} not found.")
            names = [datasets.test_datasets[self.evaluate]]
        elif self.fold == 'train':
            raise NotImplementedError("the training fold (windows, velocity / contact targets) is outside the inference path")
        else:
            raise ValueError(f"Unknown data fold: {self.fold}.")
        folder = paths.processed_datasets / ('eval' if (self.finetune or self.evaluate) else '')
        for name in names:
            try:                                                       # data.py:50-54: a bad file is reported, not fatal
                yield safe_torch_load(folder / name, self._trusted)
            except UntrustedFileError:
                raise                                                  # not a broken file: the caller must decide (trusted=True)
            except Exception as e:
                print(f"Error processing {name}: {e}.")

    def _body_fk(self, smpl):
        import os
        from .body_model import ParametricModel
        if smpl is None:
            smpl = ParametricModel(str(paths.smpl_file)) if os.path.exists(str(paths.smpl_file)) else ParametricModel.synthetic()
        elif not isinstance(smpl, ParametricModel):
            smpl = ParametricModel(data=smpl)
        self.bodymodel = smpl                                           # data.py:24
        return smpl.forward_kinematics

    # ---- one file: every sequence x every combo (data.py:57-85) -----------------------------------------------------
    def _add_file(self, fd, masks):
        for acc, ori, pose, tran in zip(fd['acc'], fd['ori'], fd['pose'], fd['tran']):
            acc = torch.as_tensor(acc).float()[:, :5] / amass.acc_scale
            ori = torch.as_tensor(ori).float()[:, :5]
            pose = torch.as_tensor(pose).float().reshape(-1, 24, 3, 3)
            tran = torch.as_tensor(tran).float()
            joint = self._fk(pose)[1].detach().cpu().reshape(-1, 24, 3)
            # all combos at once: [C,1,5,1] masks against [1,N,5,3] / [1,N,5,9]
            m = masks.reshape(-1, 1, 5, 1)
            imu = torch.cat(((acc.unsqueeze(0) * m).flatten(2), (ori.flatten(2).unsqueeze(0) * m).flatten(2)), dim=2)
            for k in range(imu.shape[0]):
                self.data['imu_inputs'].append(imu[k])
                self.data['pose_outputs'].append(pose)
                self.data['joint_outputs'].append(joint)
                self.data['tran_outputs'].append(tran)

    def __len__(self):
        return len(self.data['imu_inputs'])

    def __getitem__(self, idx):
        """(imu, pose as r6d of all 24 joints, joint, tran) -- data.py:94-102 for the evaluate / finetune folds."""
        pose = rotation_matrix_to_r6d(self.data['pose_outputs'][idx]).reshape(-1, 24 * 6)
        return (self.data['imu_inputs'][idx].float(), pose, self.data['joint_outputs'][idx].float(),
                self.data['tran_outputs'][idx].float())

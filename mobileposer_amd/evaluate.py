"""Evaluation harness: mirror of mobileposer/evaluate.py (PoseEvaluator :16-36, evaluate_pose :39-107, CLI :110-126).

Per sequence: ``model.reset()`` -> ``forward_offline`` -> (env ONLINE=1: ``forward_online`` for every frame plus
5 repeated tail frames, first 5 outputs dropped, evaluate.py:62-64) -> errors.  ``FullMotionEvaluator`` follows
articulate/evaluator.py:292-343 with forward kinematics and mesh skinning on the GPU (mp_fk_mesh); rotation angles
are 2*asin(|R_p^T R_t - I|_F / (2*sqrt 2)) instead of a per-matrix cv2.Rodrigues call (same value); the
mean / std reductions of the error table are plain torch reductions on the device.
"""
import argparse
import math
import os

import torch

from .config import datasets, joint_set
from .data import PoseDataset


def getenv(key, default=0):
    """helpers.py:4-5."""
    return type(default)(os.getenv(key, default))


def r6d_to_rotation_matrix_torch(r6d):
    """Ground-truth side only (evaluate.py:60): plain torch restatement of angular.py:167-182."""
    r6d = r6d.reshape(-1, 6)
    c0 = r6d[:, 0:3] / r6d[:, 0:3].norm(dim=1, keepdim=True)
    u = r6d[:, 3:6] - (c0 * r6d[:, 3:6]).sum(dim=1, keepdim=True) * c0
    c1 = u / u.norm(dim=1, keepdim=True)
    c2 = torch.cross(c0, c1, dim=1)
    r = torch.stack((c0, c1, c2), dim=-1)
    r[torch.isnan(r)] = 0
    return r


def angle_between(Ra, Rb):
    D = Ra.transpose(-1, -2) @ Rb
    n = (D - torch.eye(3, device=D.device)).flatten(-2).norm(dim=-1)
    return 2.0 * torch.asin((n / (2.0 * math.sqrt(2.0))).clamp(0.0, 1.0))


class FullMotionEvaluator:
    """articulate/evaluator.py:269-343: 10 x [mean, std] error table (mean shape, rotation-matrix inputs)."""

    def __init__(self, model, joint_mask=None, fps=60, align_joint=0):
        self.model, self.joint_mask, self.fps, self.align_joint = model, joint_mask, fps, align_joint

    def __call__(self, pose_p, pose_t, tran_p=None, tran_t=None):
        f, m = self.fps, self.model
        mesh = m.n_vertex > 0
        if mesh:
            Rg_p, j_p, v_p = m.forward_kinematics(pose_p, tran_p, calc_mesh=True)
            Rg_t, j_t, v_t = m.forward_kinematics(pose_t, tran_t, calc_mesh=True)
        else:
            Rg_p, j_p = m.forward_kinematics(pose_p, tran_p)
            Rg_t, j_t = m.forward_kinematics(pose_t, tran_t)
        dev = j_p.device
        pl_p = pose_p.to(dev).reshape(-1, 24, 3, 3)
        pl_t = pose_t.to(dev).reshape(-1, 24, 3, 3)
        off = (j_t[:, self.align_joint] - j_p[:, self.align_joint]).unsqueeze(1)
        je = (j_p + off - j_t).norm(dim=2)
        ve = (v_p + off - v_t).norm(dim=2) if mesh else torch.full((1, 1), float("nan"), device=dev)
        lae = torch.rad2deg(angle_between(pl_p, pl_t))
        gae = torch.rad2deg(angle_between(Rg_p, Rg_t))
        jkp = ((j_p[3:] - 3 * j_p[2:-1] + 3 * j_p[1:-2] - j_p[:-3]) * (f ** 3)).norm(dim=2)
        jkt = ((j_t[3:] - 3 * j_t[2:-1] + 3 * j_t[1:-2] - j_t[:-3]) * (f ** 3)).norm(dim=2)
        te = ((j_p[f:, :1] - j_p[:-f, :1]) - (j_t[f:, :1] - j_t[:-f, :1])).norm(dim=2) * 100
        jm = self.joint_mask
        zero = torch.zeros(1, 1, device=dev)
        rows = [je, ve, lae, gae, jkp, jkt, te,
                je[:, jm] if jm is not None else zero, lae[:, jm] if jm is not None else zero,
                gae[:, jm] if jm is not None else zero]

        def ms(x):
            if x.numel() == 0:
                return torch.full((2,), float("nan"), device=dev)
            return torch.stack((x.mean(), x.std(dim=0).mean() if x.shape[0] > 1 else x.new_tensor(float("nan"))))

        return torch.stack([ms(x) for x in rows])


class PoseEvaluator:
    """evaluate.py:16-36."""
    names = ['SIP Error (deg)', 'Angular Error (deg)', 'Masked Angular Error (deg)', 'Positional Error (cm)',
             'Masked Positional Error (cm)', 'Mesh Error (cm)', 'Jitter Error (100m/s^3)', 'Distance Error (cm)']

    def __init__(self, model, joint_mask=(2, 5, 16, 20), fps=datasets.fps):
        self.model = model
        self._eval_fn = FullMotionEvaluator(model, joint_mask=list(joint_mask), fps=fps)

    def eval(self, pose_p, pose_t, joint_p=None, tran_p=None, tran_t=None):
        dev = self.model.device
        pose_p = pose_p.clone().view(-1, 24, 3, 3).to(dev)
        pose_t = pose_t.clone().view(-1, 24, 3, 3).to(dev)
        tran_p = tran_p.clone().view(-1, 3).to(dev)
        tran_t = tran_t.clone().view(-1, 3).to(dev)
        eye = torch.eye(3, device=dev)
        pose_p[:, joint_set.ignored] = eye                                               # evaluate.py:25-26
        pose_t[:, joint_set.ignored] = eye
        errs = self._eval_fn(pose_p, pose_t, tran_p=tran_p, tran_t=tran_t)
        # evaluate.py:29
        return torch.stack([errs[9], errs[3], errs[9], errs[0] * 100, errs[7] * 100, errs[1] * 100, errs[4] / 100, errs[6]])

    @classmethod
    def print(cls, errors):
        for i, name in enumerate(cls.names):
            print('%s: %.2f (+/- %.2f)' % (name, errors[i, 0], errors[i, 1]))


@torch.no_grad()
def evaluate_pose(model, dataset, num_future_frame=5, verbose=True):
    """evaluate.py:39-107 (translation-window statistics of evaluate_tran omitted)."""
    dev = model.device
    evaluator = PoseEvaluator(model)
    offline_errs, online_errs = [], []
    model.eval()
    for imu, pose_t, _joint, tran_t in dataset:
        x = imu.to(dev)
        model.reset()
        pose_p, _joint_p, tran_p, _ = model.forward_offline(x.unsqueeze(0), [x.shape[0]])
        pose_t_m = r6d_to_rotation_matrix_torch(pose_t.to(dev)).view(-1, 24, 3, 3)
        offline_errs.append(evaluator.eval(pose_p, pose_t_m, tran_p=tran_p, tran_t=tran_t))
        if getenv("ONLINE"):
            frames = torch.cat((x, x[-1].repeat(num_future_frame, 1)))
            res = [model.forward_online(f) for f in frames]
            pose_o, _j, tran_o, _c = [torch.stack(_)[num_future_frame:] for _ in zip(*res)]
            online_errs.append(evaluator.eval(pose_o, pose_t_m, tran_p=tran_o, tran_t=tran_t))
    out = {"offline": torch.stack(offline_errs).nanmean(dim=0) if offline_errs else None}
    if verbose:
        print('============== offline ================')
        PoseEvaluator.print(out["offline"])
    if online_errs:
        out["online"] = torch.stack(online_errs).nanmean(dim=0)
        if verbose:
            print('============== online ================')
            PoseEvaluator.print(out["online"])
    return out


def synthetic_dataset(n_seq=2, frames=90, seed=0):
    """A dataset dict in the reference's on-disk format, for smoke runs without the licensed data."""
    import numpy as np
    from .synthetic import _random_rotations
    rng = np.random.Generator(np.random.PCG64(seed))
    d = {"acc": [], "ori": [], "pose": [], "tran": []}
    for _ in range(n_seq):
        d["acc"].append(torch.from_numpy((rng.standard_normal((frames, 6, 3)) * 5).astype("float32")))
        d["ori"].append(torch.from_numpy(_random_rotations(rng, frames * 6).reshape(frames, 6, 3, 3).astype("float32")))
        d["pose"].append(torch.from_numpy(_random_rotations(rng, frames * 24).reshape(frames, 24, 3, 3).astype("float32")))
        d["tran"].append(torch.from_numpy(np.cumsum(rng.standard_normal((frames, 3)) * 0.01, axis=0).astype("float32")))
    return d


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', type=str, required=True, help="weights .pth (state dict) or 'synthetic'")
    ap.add_argument('--dataset', type=str, default='dip')
    ap.add_argument('--smpl', type=str, default=None, help="SMPL pickle; default: smpl/basicmodel_m.pkl or synthetic")
    ap.add_argument('--data-dir', type=str, default='data/processed_datasets/eval')
    ap.add_argument('--max-combos', type=int, default=None)
    args = ap.parse_args(argv)
    from .model_utils import load_model
    from .net import MobilePoserNet
    from . import synthetic
    smpl_file = args.smpl or ('smpl/basicmodel_m.pkl' if os.path.exists('smpl/basicmodel_m.pkl') else None)
    if args.model == 'synthetic':
        model = MobilePoserNet(smpl_file=smpl_file)
        model.load_state_dict(synthetic.make_weights(0))
    else:
        model = load_model(args.model, smpl_file=smpl_file)
    if args.dataset == 'synthetic':
        data = synthetic_dataset()
    else:
        if args.dataset not in datasets.test_datasets:
            raise ValueError(f"Test dataset: {args.dataset} not found.")                 # evaluate.py:120-121
        data = os.path.join(args.data_dir, datasets.test_datasets[args.dataset])
    from .config import amass
    combos = dict(list(amass.combos.items())[:args.max_combos]) if args.max_combos else None
    dataset = PoseDataset(data, fk=model.forward_kinematics, combos=combos)
    print(f"Starting evaluation: {args.dataset.capitalize()}")
    evaluate_pose(model, dataset)


if __name__ == '__main__':
    main()

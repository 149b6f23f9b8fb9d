"""Evaluation harness with the reference's surface (mobileposer/evaluate.py): ``PoseEvaluator`` (:16-36),
``evaluate_pose(model, dataset, num_past_frame=20, num_future_frame=5, evaluate_tran=False)`` (:39-107) and the CLI
``python -m mobileposer_amd.evaluate --model W --dataset dip`` (:110-126, env ONLINE as in the reference).

Per sequence: ``model.reset()`` -> ``forward_offline`` -> (ONLINE=1: ``forward_online`` for every frame plus
``num_future_frame`` repeats of the last one, the first ``num_future_frame`` outputs dropped, :62-64) -> error table.
``FullMotionEvaluator`` follows articulate/evaluator.py:292-343 with forward kinematics and mesh skinning on the GPU
(mp_fk_mesh); a rotation angle is 2*asin(|R_p^T R_t - I|_F / (2*sqrt 2)) instead of a per-matrix cv2.Rodrigues call (the
same number); means / stds are torch reductions on the device.  Sequence tables are combined with ``mean`` exactly as
the reference does -- a sequence shorter than one second makes the 1-s distance row NaN there, and here.
"""
import argparse
import os

import numpy as np
import torch

from .config import datasets, joint_set
from .data import PoseDataset


def getenv(key, default=0):
    """helpers.py:4-5."""
    return type(default)(os.getenv(key, default))


class FullMotionEvaluator:
    """articulate/evaluator.py:269-343: 10 x [mean, std] error table (mean shape, rotation-matrix inputs) -- computed by
    ONE library call (mp_eval_metrics: FK + skinning of prediction and ground truth and all ten metrics on the GPU).
    ``model``: a MobilePoserNet or a ParametricModel (anything with their ``eval_metrics``) -- or, as in the reference
    (``FullMotionEvaluator(paths.smpl_file, joint_mask=..., fps=...)``, evaluate.py:18), the path of an SMPL model file, from which
    a body-only native handle is built.  ``align_joint``: a joint index (the reference takes an enum member and reads ``.value``)."""

    def __init__(self, model, joint_mask=None, fps=60, align_joint=0, ignored=(), device="cuda:0"):
        if isinstance(model, (str, os.PathLike)):
            from .body_model import ParametricModel
            model = ParametricModel(str(model), device=device)
        if joint_mask is not None and hasattr(joint_mask, "tolist"):
            joint_mask = joint_mask.tolist()
        align_joint = 0 if align_joint is None else int(getattr(align_joint, "value", align_joint))
        self.model, self.joint_mask, self.fps, self.align_joint, self.ignored = model, joint_mask, fps, align_joint, ignored

    def __call__(self, pose_p, pose_t, shape_p=None, shape_t=None, tran_p=None, tran_t=None):
        if shape_p is not None or shape_t is not None:
            raise NotImplementedError("per-sequence shapes in the evaluator (no reference caller passes them: evaluate.py:28)")
        return self.model.eval_metrics(pose_p, pose_t, tran_p, tran_t, fps=self.fps, align_joint=self.align_joint,
                                       joint_mask=self.joint_mask, ignored=self.ignored)


# The eight printed metrics (evaluate.py:29,33-35): label, row of the FullMotionEvaluator table it is taken from, unit factor.
# Row 9 = masked global angle (joints 2, 5, 16, 20) serves "SIP" and "Masked Angular", row 3 = global angle, rows 0 / 7 / 1 =
# joint / masked joint / vertex position (m -> cm), row 4 = predicted jitter (-> 100 m/s^3), row 6 = 1-s root distance error.
METRICS = (("SIP Error (deg)", 9, 1.0), ("Angular Error (deg)", 3, 1.0), ("Masked Angular Error (deg)", 9, 1.0),
           ("Positional Error (cm)", 0, 100.0), ("Masked Positional Error (cm)", 7, 100.0), ("Mesh Error (cm)", 1, 100.0),
           ("Jitter Error (100m/s^3)", 4, 0.01), ("Distance Error (cm)", 6, 1.0))


class PoseEvaluator:
    """evaluate.py:16-36.  ``model``: the MobilePoserNet whose handle runs FK / skinning (its body constants already live on the
    GPU); ``PoseEvaluator()`` as the reference writes it builds a body-only model from ``paths.smpl_file`` (evaluate.py:18)."""

    def __init__(self, model=None, joint_mask=(2, 5, 16, 20), fps=datasets.fps):
        if model is None:
            from .config import paths
            model = str(paths.smpl_file)
        # joints the network does not predict count as identity in both poses (evaluate.py:25-26): done inside the call
        self._eval_fn = FullMotionEvaluator(model, joint_mask=list(joint_mask), fps=fps, ignored=joint_set.ignored)
        self.model = self._eval_fn.model
        self._rows = torch.tensor([row for _, row, _ in METRICS])
        self._scale = torch.tensor([scale for _, _, scale in METRICS], dtype=torch.float32).unsqueeze(1)

    def eval(self, pose_p, pose_t, joint_p=None, tran_p=None, tran_t=None):
        table = self._eval_fn(pose_p, pose_t, tran_p=tran_p, tran_t=tran_t)
        return table[self._rows.to(table.device)] * self._scale.to(table.device)

    @staticmethod
    def print(errors):
        for (label, _, _), (mean, std) in zip(METRICS, errors.tolist()):
            print('%s: %.2f (+/- %.2f)' % (label, mean, std))


def distance_window_pairs(move_distance, window):
    """evaluate.py:74-83: walking ``start`` over the frames, ``end`` = the first frame (never moving backwards) by which the
    ground truth has travelled at least ``window`` metres since ``start``; a pair is kept when its ``end`` is new.
    ``move_distance``: non-decreasing cumulative path length [N] (float32, as the reference accumulates it)."""
    d = np.asarray(move_distance, dtype=np.float32)
    n = d.shape[0]
    if n < 2:
        return []
    # end(start) = first index e >= 1 with d[e] - d[start] >= window, evaluated in float32 like the reference's comparison
    ends = np.empty(n, dtype=np.int64)
    e = 1
    for s in range(n):
        while e < n and np.float32(d[e] - d[s]) < np.float32(window):
            e += 1
        ends[s] = e
        if e >= n:
            ends[s:] = n
            break
    pairs, last = [], -1
    for s in range(n):
        if ends[s] >= n:
            break
        if ends[s] != last:
            pairs.append((s, int(ends[s])))
            last = int(ends[s])
    return pairs


def translation_window_errors(tran_p, tran_t, windows=range(1, 8)):
    """evaluate.py:66-91 for one sequence: {window: mean relative distance error over the pairs} (windows without a pair
    are absent)."""
    tran_t = torch.as_tensor(tran_t).float().cpu()
    tran_p = torch.as_tensor(tran_p).float().cpu()
    step = (tran_t[1:] - tran_t[:-1]).norm(dim=1).numpy()
    dist = np.zeros(tran_t.shape[0], dtype=np.float32)
    for j in range(step.shape[0]):                        # sequential float32 accumulation (:69-70)
        dist[j + 1] = dist[j] + step[j]
    out = {}
    for w in windows:
        pairs = distance_window_pairs(dist, w)
        if not pairs:
            continue
        s = torch.tensor([p[0] for p in pairs])
        e = torch.tensor([p[1] for p in pairs])
        moved = torch.from_numpy(dist)[e] - torch.from_numpy(dist)[s]
        err = ((tran_t[e] - tran_t[s]) - (tran_p[e] - tran_p[s])).norm(dim=1) / moved * w
        out[w] = float(err.sum() / len(pairs))
    return out


@torch.no_grad()
def evaluate_pose(model, dataset, num_past_frame=20, num_future_frame=5, evaluate_tran=False, verbose=True):
    """evaluate.py:39-107.  Returns {"offline": [8,2], "online": [8,2] (ONLINE=1), "tran": [8] (evaluate_tran)} and prints
    what the reference prints."""
    dev = model.device
    evaluator = PoseEvaluator(model)
    tables = {"offline": [], "online": []}
    tran_errors = {w: [] for w in range(1, 8)}
    online = bool(getenv("ONLINE"))
    model.eval()
    for imu, pose_t, _joint, tran_t in dataset:
        x = imu.to(dev)
        model.reset()
        pose_p, _joints_p, tran_p, _contact = model.forward_offline(x.unsqueeze(0), [x.shape[0]])
        pose_gt = model.r6d_to_rotation_matrix(pose_t).view(-1, 24, 3, 3)                   # evaluate.py:60
        tables["offline"].append(evaluator.eval(pose_p, pose_gt, tran_p=tran_p, tran_t=tran_t))
        if evaluate_tran:
            for w, v in translation_window_errors(tran_p, tran_t).items():
                tran_errors[w].append(v)
        if online:
            feed = torch.cat((x, x[-1:].expand(num_future_frame, -1)))
            if hasattr(model, "forward_online_replay") and not getenv("MP_ONLINE_TICKS"):
                # the T + 5 forward_online calls of evaluate.py:62-64 as one library call (mp_stream_replay, round 5)
                pose_all, _j, tran_all, _c = model.forward_online_replay(feed)
                pose_o, tran_o = pose_all[num_future_frame:], tran_all[num_future_frame:]
            else:
                frames = [model.forward_online(f) for f in feed]
                pose_o = torch.stack([fr[0] for fr in frames])[num_future_frame:]
                tran_o = torch.stack([fr[2] for fr in frames])[num_future_frame:]
            tables["online"].append(evaluator.eval(pose_o, pose_gt, tran_p=tran_o, tran_t=tran_t))
        if hasattr(model, "finish"):
            model.finish()          # raises if a kernel of this sequence gave up a wait and was not repaired (recovery off)
    out = {}
    for name in ("offline", "online"):
        if tables[name]:
            out[name] = torch.stack(tables[name]).mean(dim=0)                       # mean, as evaluate.py:99-102
            if verbose:
                print('============== %s ================' % name)
                PoseEvaluator.print(out[name])
    if evaluate_tran:
        out["tran"] = [0] + [float(torch.tensor(v).mean()) if v else float("nan") for v in tran_errors.values()]
        if verbose:
            print(out["tran"])
    return out


def synthetic_dataset(n_seq=2, frames=90, seed=0):
    """A dataset dict in the reference's on-disk format, for smoke runs without the licensed data."""
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
    """evaluate.py:110-126: ``--model`` and ``--dataset`` only; everything else comes from config.paths, as in the reference.
    (``--model synthetic`` / ``--dataset synthetic`` are smoke-run conveniences for machines without the licensed files.)"""
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', type=str, required=True)
    ap.add_argument('--dataset', type=str, default='dip')
    ap.add_argument('--trusted', action='store_true',
                    help="unpickle --model / the dataset file in full (files you trust only; default: tensors and plain containers)")
    args = ap.parse_args(argv)
    from . import synthetic
    from .model_utils import load_model
    from .net import MobilePoserNet
    if args.model == 'synthetic':
        model = MobilePoserNet().load_state_dict(synthetic.make_weights(0))
    else:
        model = load_model(args.model, trusted=args.trusted)
    if args.dataset == 'synthetic':
        dataset = PoseDataset(fold='test', evaluate='dip', data=synthetic_dataset(), fk=model.forward_kinematics)
    else:
        if args.dataset not in datasets.test_datasets:
            raise ValueError(f"Test dataset: {args.dataset} not found.")
        dataset = PoseDataset(fold='test', evaluate=args.dataset, fk=model.forward_kinematics, trusted=args.trusted)
    print(f"Starting evaluation: {args.dataset.capitalize()}")
    evaluate_pose(model, dataset)


if __name__ == '__main__':
    main()

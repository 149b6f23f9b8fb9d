#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running THE REFERENCE ITSELF on CPU.

Runs only in the build container (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

What it does: installs a stub ``lightning`` module (the reference's model classes
derive from L.LightningModule, models/net.py:7,22), writes the synthetic SMPL pickle of
``mobileposer_amd.synthetic.synthetic_smpl`` to ``<tmp>/smpl/basicmodel_m.pkl`` (paths are
CWD-relative, config.py:28-30), imports ``mobileposer`` from /root/reference, loads the seeded
numpy weights of ``mobileposer_amd.synthetic.make_weights`` via ``load_state_dict`` and records
inputs + outputs of every hot-path function as small .npz files.  Only data is written -- no
reference source or bytecode enters the repository.
"""
import json
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import scipy.sparse
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from mobileposer_amd import synthetic  # noqa: E402

REFERENCE = "/root/reference"


def install_stub_lightning():
    L = types.ModuleType("lightning")

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        @property
        def device(self):
            return torch.device("cpu")

    class LightningDataModule:
        pass

    L.LightningModule = LightningModule
    L.LightningDataModule = LightningDataModule
    sys.modules["lightning"] = L


def to_torch_sd(sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def main():
    torch.set_num_threads(4)
    install_stub_lightning()
    work = tempfile.mkdtemp(prefix="mp_golden_")
    os.makedirs(os.path.join(work, "smpl"))
    smpl = synthetic.synthetic_smpl()
    pk = dict(smpl)
    pk["J_regressor"] = scipy.sparse.csc_matrix(smpl["J_regressor"])
    with open(os.path.join(work, "smpl", "basicmodel_m.pkl"), "wb") as f:
        pickle.dump(pk, f)
    os.chdir(work)
    sys.path.insert(0, REFERENCE)
    sys.dont_write_bytecode = True
    from mobileposer.models import MobilePoserNet           # noqa: E402
    from mobileposer.config import paths                    # noqa: E402
    import mobileposer.articulate as art                    # noqa: E402

    def new_model(seed=0):
        m = MobilePoserNet()
        m.load_state_dict(to_torch_sd(synthetic.make_weights(seed)))
        m.eval()
        return m

    out = {}

    # ---- G7 manifest ----------------------------------------------------------------------------
    net = new_model()
    manifest = [[k, list(v.shape)] for k, v in net.state_dict().items()]
    with open(os.path.join(HERE, "g7_manifest.json"), "w") as f:
        json.dump({"keys": manifest, "parent": [(-1 if p is None else int(p)) for p in net.bodymodel.parent],
                   "floor_y": net.floor_y, "feet_pos": net.feet_pos.numpy().tolist()}, f, indent=0)

    # ---- G1 per-module RNN, ragged lengths, with and without initial state ------------------------
    lengths = [17, 9, 13, 17]
    g1 = {"lengths": np.array(lengths)}
    with torch.no_grad():
        for name, mod, n_in in (("joints", net.joints.joints, 60), ("pose", net.pose.pose, 132),
                                ("foot_contact", net.foot_contact.footcontact, 132), ("velocity", net.velocity.vel, 132)):
            rng = np.random.Generator(np.random.PCG64(100 + n_in + len(name)))
            x = rng.standard_normal((4, 17, n_in)).astype(np.float32) * 0.5
            y, ol, (h, c) = mod(torch.from_numpy(x), lengths)
            g1[f"{name}_x"], g1[f"{name}_y"], g1[f"{name}_h"], g1[f"{name}_c"] = x, y.numpy(), h.numpy(), c.numpy()
            y2, _, (h2, c2) = mod(torch.from_numpy(x), lengths, (h, c))          # carry state in (velocity.py:47)
            g1[f"{name}_y2"], g1[f"{name}_h2"], g1[f"{name}_c2"] = y2.numpy(), h2.numpy(), c2.numpy()
    np.savez_compressed(os.path.join(HERE, "g1_rnn.npz"), **g1)

    # ---- G2 full forward: equal lengths and ragged ------------------------------------------------
    imu = synthetic.make_imu(3, 25, seed=11)
    g2 = {"imu": imu}
    with torch.no_grad():
        for tag, lens in (("eq", [25, 25, 25]), ("rag", [25, 11, 18])):
            m = new_model()
            pose, joints, vel, contact = m.forward(torch.from_numpy(imu), lens)
            g2[f"{tag}_lengths"] = np.array(lens)
            g2[f"{tag}_pose"], g2[f"{tag}_joints"] = pose.numpy(), joints.numpy()
            g2[f"{tag}_vel"], g2[f"{tag}_contact"] = vel.numpy(), contact.numpy()
            h, c = m.velocity.rnn_state
            g2[f"{tag}_vel_h"], g2[f"{tag}_vel_c"] = h.numpy(), c.numpy()
            # raw 96-d r6d too (the quantity before Gram-Schmidt amplification)
            x132 = torch.cat((joints, torch.from_numpy(imu)[:, :joints.shape[1]]), dim=-1)
            g2[f"{tag}_r6d"] = m.pose(x132, lens).numpy()
    np.savez_compressed(os.path.join(HERE, "g2_forward.npz"), **g2)

    # ---- G3 r6d -> R -> full -> local, with degenerate rows (Q8) ----------------------------------
    rng = np.random.Generator(np.random.PCG64(3))
    r6d = rng.standard_normal((64, 96)).astype(np.float32)
    r6d[5, 0:6] = 0.0                                   # zero vectors  -> NaN -> 0
    r6d[6, 6:12] = [1, 2, 3, 2, 4, 6]                   # colinear      -> NaN -> 0 in columns 1,2
    r6d[7, 12:18] = [0, 0, 0, 1, 0, 0]                  # zero first vector
    with torch.no_grad():
        full = net._reduced_global_to_full(torch.from_numpy(r6d))
        rot = art.math.r6d_to_rotation_matrix(torch.from_numpy(r6d))
    np.savez_compressed(os.path.join(HERE, "g3_r6d_ik.npz"), r6d=r6d, pose=full.numpy(), rot=rot.numpy())

    # ---- G4 forward_offline: two consecutive sequences, stale velocity state (Q1) -----------------
    T = 200
    imu_a = synthetic.make_imu(1, T, seed=21)
    imu_b = synthetic.make_imu(1, T, seed=22)
    g4 = {"imu_a": imu_a, "imu_b": imu_b}
    with torch.no_grad():
        m = new_model()
        for tag, x in (("a", imu_a), ("b", imu_b), ("a_again", imu_a)):
            m.reset()
            pose, joints, tran, contact = m.forward_offline(torch.from_numpy(x), [T])
            g4[f"{tag}_pose"], g4[f"{tag}_joints"] = pose.numpy(), joints.numpy()
            g4[f"{tag}_tran"], g4[f"{tag}_contact"] = tran.numpy(), contact.numpy()
        m.reset()
        m.velocity.rnn_state = None                      # explicit clear restores the first answer
        pose, joints, tran, contact = m.forward_offline(torch.from_numpy(imu_a), [T])
        g4["a_cleared_tran"] = tran.numpy()
    np.savez_compressed(os.path.join(HERE, "g4_offline.npz"), **g4)

    # ---- G5 forward_online: 60 consecutive frames from reset() (Q5, Q6) ---------------------------
    n_on = 60
    imu_on = synthetic.make_imu(1, n_on, seed=31)[0]
    poses, jnts, trans, cons = [], [], [], []
    with torch.no_grad():
        m = new_model()
        m.reset()
        for f in torch.from_numpy(imu_on):
            p, j, t, c = m.forward_online(f)
            poses.append(p.numpy()); jnts.append(j.numpy()[40]); trans.append(t.numpy()); cons.append(c.numpy())
        h, c = m.velocity.rnn_state
    np.savez_compressed(os.path.join(HERE, "g5_online.npz"), imu=imu_on, pose=np.stack(poses), joints40=np.stack(jnts),
                        tran=np.stack(trans), contact=np.stack(cons), vel_h=h.numpy(), vel_c=c.numpy(),
                        current_root_y=np.float64(m.current_root_y))

    # ---- G6 forward kinematics (joints, and the LBS mesh used by the evaluator) -------------------
    rng = np.random.Generator(np.random.PCG64(6))
    pose = synthetic._random_rotations(rng, 32 * 24).reshape(32, 24, 3, 3).astype(np.float32)
    tran = rng.standard_normal((32, 3)).astype(np.float32)
    with torch.no_grad():
        Rg, jg = net.bodymodel.forward_kinematics(torch.from_numpy(pose))
        Rg2, jg2, vg2 = net.bodymodel.forward_kinematics(torch.from_numpy(pose), tran=torch.from_numpy(tran), calc_mesh=True)
    np.savez_compressed(os.path.join(HERE, "g6_fk.npz"), pose=pose, tran=tran, R_global=Rg.numpy(), joint=jg.numpy(),
                        joint_tran=jg2.numpy(), vert_tran=vg2.numpy())
    # ---- G8 PoseDataset eval-mode input formation (data.py:57-85,94-107) on a synthetic dip_test.pt -------
    rng = np.random.Generator(np.random.PCG64(8))
    seqs = []
    for n in (30, 20):
        seqs.append({
            "acc": (rng.standard_normal((n, 6, 3)) * 5.0).astype(np.float32),
            "ori": synthetic._random_rotations(rng, n * 6).reshape(n, 6, 3, 3).astype(np.float32),
            "pose": synthetic._random_rotations(rng, n * 24).reshape(n, 24, 3, 3).astype(np.float32),
            "tran": rng.standard_normal((n, 3)).astype(np.float32),
        })
    os.makedirs(os.path.join(work, "data", "processed_datasets", "eval"))
    torch.save({k: [torch.from_numpy(sq[k]) for sq in seqs] for k in ("acc", "ori", "pose", "tran")},
               os.path.join(work, "data", "processed_datasets", "eval", "dip_test.pt"))
    from mobileposer.data import PoseDataset            # noqa: E402
    ds = PoseDataset(fold="test", evaluate="dip")
    g8 = {"n_items": np.int64(len(ds))}
    for k, sq in enumerate(seqs):
        for name, v in sq.items():
            g8[f"in{k}_{name}"] = v
    for idx in range(len(ds)):
        imu, pose, joint, tran = ds[idx]
        g8[f"item{idx}_imu"], g8[f"item{idx}_pose"] = imu.numpy(), pose.numpy()
        g8[f"item{idx}_joint"], g8[f"item{idx}_tran"] = joint.numpy(), tran.numpy()
    np.savez_compressed(os.path.join(HERE, "g8_dataset.npz"), **g8)

    # ---- G9 the evaluator (evaluate.py:16-29 PoseEvaluator.eval -> articulate/evaluator.py:292-343) ------------
    # cv2 is not installed here: rotation_matrix_to_axis_angle (angular.py:161-164) gets an oracle-side stand-in
    # with the same contract (Rodrigues vector of a rotation matrix, here via scipy); only its norm is used.
    cv2 = types.ModuleType("cv2")

    def _rodrigues(R):
        from scipy.spatial.transform import Rotation
        return Rotation.from_matrix(np.asarray(R, dtype=np.float64)).as_rotvec().reshape(3, 1).astype(np.float32), None

    cv2.Rodrigues = _rodrigues
    sys.modules["cv2"] = cv2
    rng = np.random.Generator(np.random.PCG64(9))
    n = 45
    pose_t = synthetic._random_rotations(rng, n * 24).reshape(n, 24, 3, 3)
    noise = rng.standard_normal((n, 24, 3)) * 0.15
    from scipy.spatial.transform import Rotation
    pose_p = np.einsum("njab,njbc->njac", pose_t, Rotation.from_rotvec(noise.reshape(-1, 3)).as_matrix().reshape(n, 24, 3, 3))
    tran_t = np.cumsum(rng.standard_normal((n, 3)) * 0.02, axis=0)
    tran_p = tran_t + np.cumsum(rng.standard_normal((n, 3)) * 0.01, axis=0)
    ev = art.FullMotionEvaluator(str(paths.smpl_file), joint_mask=torch.tensor([2, 5, 16, 20]), fps=30)
    pp, pt = torch.from_numpy(pose_p).float(), torch.from_numpy(pose_t).float()
    ign = [0, 7, 8, 10, 11, 20, 21, 22, 23]
    pp[:, ign] = torch.eye(3)
    pt[:, ign] = torch.eye(3)
    errs = ev(pp, pt, tran_p=torch.from_numpy(tran_p).float(), tran_t=torch.from_numpy(tran_t).float())
    np.savez_compressed(os.path.join(HERE, "g9_evaluator.npz"), pose_p=pose_p.astype(np.float32), pose_t=pose_t.astype(np.float32),
                        tran_p=tran_p.astype(np.float32), tran_t=tran_t.astype(np.float32), errs=errs.numpy())

    # ---- G10 live front-end math (live_demo.py:161-174 calibration, :213-236 frame formation) -------------------
    # the demo script is not importable (everything sits under __main__); the same expressions are evaluated here
    # with the reference's own articulate.math functions on seeded sensor readings
    q2r = art.math.quaternion_to_rotation_matrix
    rng = np.random.Generator(np.random.PCG64(10))
    ref_q = torch.from_numpy(rng.standard_normal(4)).float()
    tq = torch.from_numpy(rng.standard_normal((5, 4))).float()
    ta = torch.from_numpy(rng.standard_normal((5, 3)) * 9.8).float()
    smpl2imu = q2r(ref_q).view(3, 3).t()
    device2bone = smpl2imu.matmul(q2r(tq)).transpose(1, 2).matmul(torch.eye(3))
    acc_offsets = smpl2imu.matmul(ta.unsqueeze(-1))
    fq = torch.from_numpy(rng.standard_normal((7, 5, 4))).float()
    fa = torch.from_numpy(rng.standard_normal((7, 5, 3)) * 9.8).float()
    ori_raw = q2r(fq).view(-1, 5, 3, 3)
    glb_acc = (smpl2imu.matmul(fa.view(-1, 5, 3, 1)) - acc_offsets).view(-1, 5, 3)
    glb_ori = smpl2imu.matmul(ori_raw).matmul(device2bone)
    _acc = glb_acc.view(-1, 5, 3)[:, [1, 4, 3, 0, 2]] / 30
    _ori = glb_ori.view(-1, 5, 3, 3)[:, [1, 4, 3, 0, 2]]
    acc, ori = torch.zeros_like(_acc), torch.zeros_like(_ori)
    cc = [0, 3]                                           # 'lw_rp'
    acc[:, cc], ori[:, cc] = _acc[:, cc], _ori[:, cc]
    imu_input = torch.cat([acc.flatten(1), ori.flatten(1)], dim=1)
    rot = synthetic._random_rotations(rng, 24).astype(np.float32)
    aa = art.math.rotation_matrix_to_axis_angle(torch.from_numpy(rot))       # cv2 stand-in from G9
    np.savez_compressed(os.path.join(HERE, "g10_live.npz"), ref_q=ref_q.numpy(), tq=tq.numpy(), ta=ta.numpy(),
                        fq=fq.numpy(), fa=fa.numpy(), smpl2imu=smpl2imu.numpy(), device2bone=device2bone.numpy(),
                        acc_offsets=acc_offsets.numpy(), imu_input=imu_input.numpy(), rot=rot, axis_angle=aa.numpy())

    # ---- G11 evaluate_pose incl. the translation-window statistics (evaluate.py:39-107, evaluate_tran=True) --------
    # the reference's own evaluate_pose, driven with a stand-in model that returns canned predictions (what is pinned here
    # is the harness: per-sequence loop, error table aggregation with mean(), the evaluate_tran pair search and averages)
    import builtins
    import mobileposer.evaluate as ref_eval            # noqa: E402
    rng = np.random.Generator(np.random.PCG64(11))
    seqs11 = []
    for n in (150, 90, 20):                            # the 20-frame one is shorter than fps: its 1-s error row is NaN
        pose_t = synthetic._random_rotations(rng, n * 24).reshape(n, 24, 3, 3).astype(np.float32)
        noise = rng.standard_normal((n, 24, 3)) * 0.1
        pose_p = np.einsum("njab,njbc->njac", pose_t, Rotation.from_rotvec(noise.reshape(-1, 3)).as_matrix()
                           .reshape(n, 24, 3, 3)).astype(np.float32)
        tran_t = np.cumsum(np.abs(rng.standard_normal((n, 3))) * np.array([0.06, 0.002, 0.05]), axis=0).astype(np.float32)
        tran_p = (tran_t + np.cumsum(rng.standard_normal((n, 3)) * 0.004, axis=0)).astype(np.float32)
        imu = rng.standard_normal((n, 60)).astype(np.float32)
        seqs11.append(dict(imu=imu, pose_t=pose_t, pose_p=pose_p, tran_t=tran_t, tran_p=tran_p))

    class _CannedModel:
        def __init__(self):
            self.k = -1

        def eval(self):
            return self

        def reset(self):
            self.k += 1

        def forward_offline(self, x, lengths):
            sq = seqs11[self.k]
            return (torch.from_numpy(sq["pose_p"]), torch.zeros(1, x.shape[1], 72), torch.from_numpy(sq["tran_p"]),
                    torch.zeros(x.shape[1], 2))

    dataset11 = [(torch.from_numpy(sq["imu"]), art.math.rotation_matrix_to_r6d(torch.from_numpy(sq["pose_t"])).reshape(-1, 144),
                  torch.zeros(sq["imu"].shape[0], 24, 3), torch.from_numpy(sq["tran_t"])) for sq in seqs11]
    captured = {}
    ref_eval.PoseEvaluator.print = staticmethod(lambda errors: captured.setdefault("table", errors.clone()))
    real_print = builtins.print

    def _capture_print(*a, **k):
        if len(a) == 1 and isinstance(a[0], list):
            captured["tran"] = [float(v) for v in a[0]]
        else:
            real_print(*a, **k)

    builtins.print = _capture_print
    try:
        ref_eval.evaluate_pose(_CannedModel(), dataset11, evaluate_tran=True)
    finally:
        builtins.print = real_print
    g11 = {"n_seq": np.int64(len(seqs11)), "table": captured["table"].numpy(), "tran_errors": np.array(captured["tran"], dtype=np.float64)}
    for k, sq in enumerate(seqs11):
        for name, v in sq.items():
            g11[f"s{k}_{name}"] = v
    np.savez_compressed(os.path.join(HERE, "g11_evaluate.npz"), **g11)

    # ---- G13 evaluate_pose with ONLINE=1 on the real network (evaluate.py:57-65,96-103) ---------------------------
    # offline and online tables of the reference's own loop: forward_offline per sequence, then forward_online frame by
    # frame over the sequence padded with num_future_frame copies of its last frame, the first 5 outputs dropped
    rng = np.random.Generator(np.random.PCG64(13))
    seqs13 = []
    for k, n in enumerate((70, 45)):
        pose_t = synthetic._random_rotations(rng, n * 24).reshape(n, 24, 3, 3).astype(np.float32)
        tran_t = np.cumsum(rng.standard_normal((n, 3)) * 0.01, axis=0).astype(np.float32)
        seqs13.append(dict(imu=synthetic.make_imu(1, n, seed=130 + k)[0], pose_t=pose_t, tran_t=tran_t))
    dataset13 = [(torch.from_numpy(sq["imu"]), art.math.rotation_matrix_to_r6d(torch.from_numpy(sq["pose_t"])).reshape(-1, 144),
                  torch.zeros(sq["imu"].shape[0], 24, 3), torch.from_numpy(sq["tran_t"])) for sq in seqs13]
    tables13 = []
    ref_eval.PoseEvaluator.print = staticmethod(lambda errors: tables13.append(errors.clone()))
    os.environ["ONLINE"] = "1"
    try:
        ref_eval.evaluate_pose(new_model(), dataset13)
    finally:
        del os.environ["ONLINE"]
    assert len(tables13) == 2
    g13 = {"n_seq": np.int64(len(seqs13)), "offline": tables13[0].numpy(), "online": tables13[1].numpy()}
    for k, sq in enumerate(seqs13):
        for name, v in sq.items():
            g13[f"s{k}_{name}"] = v
    np.savez_compressed(os.path.join(HERE, "g13_evaluate_online.npz"), **g13)

    # ---- G12 forward kinematics with shape blending (articulate/model.py:84-89,208-240) -------------------------
    rng = np.random.Generator(np.random.PCG64(12))
    n12 = 9
    pose12 = synthetic._random_rotations(rng, n12 * 24).reshape(n12, 24, 3, 3).astype(np.float32)
    tran12 = rng.standard_normal((n12, 3)).astype(np.float32)
    g12 = {"pose": pose12, "tran": tran12}
    with torch.no_grad():
        for tag, shp in (("one", (rng.standard_normal(10) * 1.5).astype(np.float32)),
                         ("per", (rng.standard_normal((n12, 10)) * 1.5).astype(np.float32))):
            Rg, jg, vg = net.bodymodel.forward_kinematics(torch.from_numpy(pose12), shape=torch.from_numpy(shp),
                                                          tran=torch.from_numpy(tran12), calc_mesh=True)
            _, jg0 = net.bodymodel.forward_kinematics(torch.from_numpy(pose12), shape=torch.from_numpy(shp))
            g12[f"{tag}_shape"], g12[f"{tag}_R"], g12[f"{tag}_joint"] = shp, Rg.numpy(), jg.numpy()
            g12[f"{tag}_vert"], g12[f"{tag}_joint_notran"] = vg.numpy(), jg0.numpy()
    np.savez_compressed(os.path.join(HERE, "g12_fk_shape.npz"), **g12)


    # ---- G14 "trained-like" weights (round 4): make_weights(0, profile="trained") -- LSTM weights x 3, forget-gate bias + 1,
    # linear1 x 2: saturated gates, recurrent gain > 1, long memory -- the regime evaluate.py:56 feeds its 12 combos through
    def model_from(sd):
        m = MobilePoserNet()
        m.load_state_dict(to_torch_sd(sd))
        m.eval()
        return m

    sd_tr = synthetic.make_weights(0, profile="trained")
    combos6 = ["lw_rp_h", "rw_lp", "lp_h", "rp", "lw_lp", "rw_rp_h"]
    imu14 = synthetic.make_imu(6, 60, seed=141, combo=combos6)
    lens14 = [60, 31, 47, 60, 12, 55]
    g14 = {"imu": imu14, "lengths": np.array(lens14)}
    with torch.no_grad():
        m = model_from(sd_tr)
        pose, joints, vel, contact = m.forward(torch.from_numpy(imu14), lens14)
        g14["pose"], g14["joints"], g14["vel"], g14["contact"] = pose.numpy(), joints.numpy(), vel.numpy(), contact.numpy()
        h, c = m.velocity.rnn_state
        g14["vel_h"], g14["vel_c"] = h.numpy(), c.numpy()
        x132 = torch.cat((joints, torch.from_numpy(imu14)), dim=-1)
        g14["r6d"] = m.pose(x132, lens14).numpy()
        # forward_offline, T = 600: tran[i] = velocity[:i+1].sum(0) over 600 fp32 terms (net.py:154)
        imu_long = synthetic.make_imu(1, 600, seed=142, combo="lw_rp_h")
        m = model_from(sd_tr)
        m.reset()
        pose, joints, tran, contact = m.forward_offline(torch.from_numpy(imu_long), [600])
        g14["off_imu"], g14["off_pose"], g14["off_joints"] = imu_long, pose.numpy(), joints.numpy()
        g14["off_tran"], g14["off_contact"] = tran.numpy(), contact.numpy()
        # forward_online x 50 from reset()
        imu_on14 = synthetic.make_imu(1, 50, seed=143, combo="rw_lp")[0]
        m = model_from(sd_tr)
        m.reset()
        po, jo, to, co = [], [], [], []
        for f in torch.from_numpy(imu_on14):
            p_, j_, t_, c_ = m.forward_online(f)
            po.append(p_.numpy()); jo.append(j_.numpy()[40]); to.append(t_.numpy()); co.append(c_.numpy())
        h, c = m.velocity.rnn_state
        g14["on_imu"], g14["on_pose"], g14["on_joints40"] = imu_on14, np.stack(po), np.stack(jo)
        g14["on_tran"], g14["on_contact"] = np.stack(to), np.stack(co)
        g14["on_vel_h"], g14["on_vel_c"] = h.numpy(), c.numpy()
        # one batch whose row k uses combo k of config.py:60-73 (data.py:69-76), trained profile and a second init-scale seed
        from mobileposer.config import amass as ref_amass
        names = list(ref_amass.combos)
        assert names == list(synthetic.amass.combos) and all(ref_amass.combos[k] == synthetic.amass.combos[k] for k in names)
        imu12 = synthetic.make_imu(12, 40, seed=144, combo=names)
        g14["c12_imu"] = imu12
        for tag, sd12 in (("tr", sd_tr), ("s1", synthetic.make_weights(1))):
            m = model_from(sd12)
            pose, joints, vel, contact = m.forward(torch.from_numpy(imu12), [40] * 12)
            x132 = torch.cat((joints, torch.from_numpy(imu12)), dim=-1)
            g14[f"c12_{tag}_joints"], g14[f"c12_{tag}_vel"] = joints.numpy(), vel.numpy()
            g14[f"c12_{tag}_contact"], g14[f"c12_{tag}_r6d"] = contact.numpy(), m.pose(x132, [40] * 12).numpy()
    np.savez_compressed(os.path.join(HERE, "g14_trained.npz"), **g14)

    # ---- G15 the real mesh size: 6890 vertices (articulate/model.py:77-92,208-240, evaluator.py:319-322) -------------
    # synthetic_smpl(n_vertex=6890): 26 full 256-vertex chunks + a 234-vertex tail; zero-pose body of a shape, FK + skinning
    # without / with shape, and with pose blend shapes (use_pose_blendshape=True, model.py:236-238)
    smpl_big = synthetic.synthetic_smpl(n_vertex=6890)
    pkb = dict(smpl_big)
    pkb["J_regressor"] = scipy.sparse.csc_matrix(smpl_big["J_regressor"])
    big_path = os.path.join(work, "smpl", "big_m.pkl")
    with open(big_path, "wb") as f:
        pickle.dump(pkb, f)
    rng = np.random.Generator(np.random.PCG64(15))
    n15 = 3
    pose15 = synthetic._random_rotations(rng, n15 * 24).reshape(n15, 24, 3, 3).astype(np.float32)
    tran15 = rng.standard_normal((n15, 3)).astype(np.float32)
    shape15 = (rng.standard_normal((n15, 10)) * 1.5).astype(np.float32)
    g15 = {"pose": pose15, "tran": tran15, "shape": shape15}
    with torch.no_grad():
        bm = art.ParametricModel(big_path)
        _, jg, vg = bm.forward_kinematics(torch.from_numpy(pose15), tran=torch.from_numpy(tran15), calc_mesh=True)
        g15["joint"], g15["vert"] = jg.numpy(), vg.numpy()
        _, jg, vg = bm.forward_kinematics(torch.from_numpy(pose15), shape=torch.from_numpy(shape15),
                                          tran=torch.from_numpy(tran15), calc_mesh=True)
        g15["shape_joint"], g15["shape_vert"] = jg.numpy(), vg.numpy()
        j0, v0 = bm.get_zero_pose_joint_and_vertex(torch.from_numpy(shape15[:2]))
        g15["zero_joint"], g15["zero_vert"] = j0.numpy(), v0.numpy()
        bmp = art.ParametricModel(big_path, use_pose_blendshape=True)
        _, jg, vg = bmp.forward_kinematics(torch.from_numpy(pose15), shape=torch.from_numpy(shape15[:1]),
                                           tran=torch.from_numpy(tran15), calc_mesh=True)
        g15["blend_joint"], g15["blend_vert"] = jg.numpy(), vg.numpy()
        _, _, vg = bmp.forward_kinematics(torch.from_numpy(pose15), calc_mesh=True)
        g15["blend_vert_noshape"] = vg.numpy()
    np.savez_compressed(os.path.join(HERE, "g15_mesh6890.npz"), **g15)

    # ---- G16 (round 4) corner semantics recorded from the reference instead of argued: (a) ONE NaN in one IMU sample of one
    # sequence of a batch (F.relu keeps NaN, rnn.py:22; the NaN -> 0 rule of r6d_to_rotation_matrix, angular.py:181); (b) five
    # consecutive forward() calls of ONE frame each, the velocity state carried from call to call (velocity.py:45-48)
    g16 = {}
    with torch.no_grad():
        imu16 = synthetic.make_imu(5, 24, seed=161)
        bad16 = imu16.copy()
        bad16[2, 9, 7] = np.nan
        g16["nan_imu"] = bad16
        m = model_from(synthetic.make_weights(0))
        pose, joints, vel, contact = m.forward(torch.from_numpy(bad16), [24] * 5)
        g16["nan_pose"], g16["nan_joints"], g16["nan_vel"], g16["nan_contact"] = pose.numpy(), joints.numpy(), vel.numpy(), contact.numpy()
        m = model_from(sd_tr)
        one16 = synthetic.make_imu(4, 5, seed=162)                    # call k uses frame k of each of the 4 sequences
        g16["one_imu"] = one16
        for k in range(5):
            pose, joints, vel, contact = m.forward(torch.from_numpy(one16[:, k:k + 1]), [1] * 4)
            g16[f"one_joints{k}"], g16[f"one_vel{k}"], g16[f"one_contact{k}"] = joints.numpy(), vel.numpy(), contact.numpy()
        h, c = m.velocity.rnn_state
        g16["one_vel_h"], g16["one_vel_c"] = h.numpy(), c.numpy()
    np.savez_compressed(os.path.join(HERE, "g16_corners.npz"), **g16)

    # ---- G17 (round 6) the reference's OWN call shape with trained-regime weights: evaluate.py:54-58 calls forward_offline
    # (net.py:121-155) on ONE sequence of thousands of frames.  Three lengths x three input seeds (combos vary).  At this length
    # the trained-regime net amplifies rounding until two fp32 evaluations differ by 1e-4 ... 1e-3 somewhere, so beside the
    # reference's outputs (every 8th frame of r6d / joints / velocity, contact and translation in full) the file records how far
    # each member of an ENSEMBLE of fp32 evaluations is from the float64 result over ALL frames: the reference itself (torch
    # CPU), the numpy oracle, and the oracle with fourteen permuted summation orders (oracle/ensemble.py) -- the band a kernel with
    # yet another summation order has to stay inside (tests/test_gpu_round6.py).
    from oracle import ensemble as ENS
    N_PERM = 14          # 16 members in all (the first version of this golden had 5: the maximum over 5 draws of a heavy-tailed
    #                      distance is no envelope -- two of 90 checks of kernels AT the ensemble's level fell outside twice of it)
    g17 = {"lengths": np.array([2000, 2500, 3000]), "seeds": np.array([171, 172, 173]), "stride": np.array(8),
           "members": np.array(["reference", "oracle"] + ["perm%d" % k for k in range(N_PERM)]), "outputs": np.array(ENS.OUTPUTS)}
    combos17 = {171: "lw_rp_h", 172: "rw_lp", 173: "lp_h"}
    g17["combos"] = np.array([combos17[s] for s in (171, 172, 173)])
    with torch.no_grad():
        for T in (2000, 2500, 3000):
            for seed in (171, 172, 173):
                imu17 = synthetic.make_imu(1, T, seed=seed, combo=combos17[seed])
                m = model_from(sd_tr)
                m.reset()
                pose, joints, tran, contact = m.forward_offline(torch.from_numpy(imu17), [T])
                m2 = model_from(sd_tr)                                   # (forward_offline keeps neither vel nor r6d: once more)
                _, joints2, vel, _ = m2.forward(torch.from_numpy(imu17), [T])
                assert torch.equal(joints, joints2)
                r6d = m2.pose(torch.cat((joints2, torch.from_numpy(imu17)), dim=-1), [T])
                ref = {"r6d": r6d.numpy().reshape(T, 96), "joints": joints.numpy().reshape(T, 72), "vel": vel.numpy().reshape(T, 72),
                       "contact": contact.numpy().reshape(T, 2), "tran": tran.numpy().reshape(T, 3)}
                truth = ENS.offline_outputs(sd_tr, smpl["J"], imu17, T, dtype=np.float64)
                stats = [ENS.distance(ref, truth), ENS.distance(ENS.offline_outputs(sd_tr, smpl["J"], imu17, T), truth)]
                stats += [ENS.distance(ENS.offline_outputs(sd_tr, smpl["J"], imu17, T, perm_seed=1700 + k), truth) for k in range(N_PERM)]
                tag = "T%d_s%d" % (T, seed)
                g17[tag + "_imu_sum"] = np.array(imu17.astype(np.float64).sum())              # the input is regenerated from its seed
                for k in ("r6d", "joints", "vel"):
                    g17[tag + "_" + k] = ref[k][(T - 1) % 8::8].copy()                         # every 8th frame, the last one included
                g17[tag + "_contact"], g17[tag + "_tran"] = ref["contact"], ref["tran"]
                # [member][output][max, mean] of |x - float64| over all frames
                g17[tag + "_dist"] = np.array([[st[k] for k in ENS.OUTPUTS] for st in stats], dtype=np.float64)
                print("G17 %s: max |x - f64| r6d %s" % (tag, " ".join("%.1e" % st["r6d"][0] for st in stats)), flush=True)
    np.savez_compressed(os.path.join(HERE, "g17_single_sequence.npz"), **g17)

    # ---- G18 (round 6) input_lengths=None, the time-major quirk (rnn.py:15 builds nn.LSTM without batch_first, only the
    # packed path is batch-first): forward twice on [3,25,60] (the carried velocity state has batch 25) and forward_offline on
    # [1,40,60] -- 40 one-step sequences, then the solver
    g18 = {"imu": synthetic.make_imu(3, 25, seed=181), "imu1": synthetic.make_imu(1, 40, seed=182)}
    with torch.no_grad():
        m = new_model()
        for call in (0, 1):
            pose, joints, vel, contact = m.forward(torch.from_numpy(g18["imu"]), None)
            g18[f"c{call}_pose"], g18[f"c{call}_joints"] = pose.numpy(), joints.numpy()
            g18[f"c{call}_vel"], g18[f"c{call}_contact"] = vel.numpy(), contact.numpy()
        h, c = m.velocity.rnn_state
        g18["vel_h"], g18["vel_c"] = h.numpy(), c.numpy()
        m = new_model()
        m.reset()
        pose, joints, tran, contact = m.forward_offline(torch.from_numpy(g18["imu1"]), None)
        g18["off_pose"], g18["off_joints"], g18["off_tran"], g18["off_contact"] = pose.numpy(), joints.numpy(), tran.numpy(), contact.numpy()
    np.savez_compressed(os.path.join(HERE, "g18_lengths_none.npz"), **g18)

    # ---- G19 (round 6) the sub-modules called the way the reference's own code calls them (net.py:103-117): model.joints(x, l),
    # model.pose(x, l), model.foot_contact(x, l), model.velocity(x, l) (zero state), model.velocity.forward_online(x, l) twice (the
    # carried state) -- and each with input_lengths=None, where dim 0 is time (rnn.py:15,25)
    g19 = {"lengths": np.array([11, 6, 11])}
    with torch.no_grad():
        m = new_model()
        l19 = g19["lengths"].tolist()
        for name, mod, n_in in (("joints", m.joints, 60), ("pose", m.pose, 132), ("foot_contact", m.foot_contact, 132),
                                ("velocity", m.velocity, 132)):
            rng = np.random.Generator(np.random.PCG64(190 + n_in + len(name)))
            x = rng.standard_normal((3, 11, n_in)).astype(np.float32) * 0.5
            g19[f"{name}_x"] = x
            g19[f"{name}_y"] = mod(torch.from_numpy(x), l19).numpy()
            g19[f"{name}_y_none"] = mod(torch.from_numpy(x)).numpy()
        xv = torch.from_numpy(g19["velocity_x"])
        m.velocity.rnn_state = None
        for call in (0, 1):
            g19[f"online{call}"] = m.velocity.forward_online(xv, l19).numpy()
        g19["online_h"], g19["online_c"] = (t.numpy() for t in m.velocity.rnn_state)
        m.velocity.rnn_state = None
        for call in (0, 1):
            g19[f"online_none{call}"] = m.velocity.forward_online(xv).numpy()
        g19["online_none_h"], g19["online_none_c"] = (t.numpy() for t in m.velocity.rnn_state)
    np.savez_compressed(os.path.join(HERE, "g19_submodules.npz"), **g19)

    # ---- G20 (round 6) the rotation kinematics of the body model on their own (articulate/model.py:126-164): forward_kinematics_R,
    # inverse_kinematics_R -- what MobilePoserNet.global_to_local_pose is bound to (net.py:38) -- on random rotations, and the round trip
    rng = np.random.Generator(np.random.PCG64(200))
    q = rng.standard_normal((37, 24, 4))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R20 = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                    2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                    2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=-1).reshape(37, 24, 3, 3).astype(np.float32)
    with torch.no_grad():
        m = new_model()
        Rg = m.bodymodel.forward_kinematics_R(torch.from_numpy(R20))
        Rl = m.global_to_local_pose(torch.from_numpy(R20))
        back = m.bodymodel.inverse_kinematics_R(Rg)
    np.savez_compressed(os.path.join(HERE, "g20_rotation_kinematics.npz"), R=R20, fk_R=Rg.numpy(), ik_R=Rl.numpy(), ik_of_fk=back.numpy())

    # ---- G21 (round 6) the call surface of the path: parameter names and defaults of the reference's callables that a
    # caller of the hot path touches, as inspect.signature reports them -- names and literals only, no source text
    import inspect
    from mobileposer.models import Joints, Poser, FootContact, Velocity   # noqa: E402
    from mobileposer.utils.model_utils import load_model as ref_load_model     # noqa: E402

    def sig(fn):
        out = []
        for name, prm in inspect.signature(fn).parameters.items():
            if name == "self":
                continue
            d = prm.default
            out.append([name, None if d is inspect.Parameter.empty else repr(d) if isinstance(d, (int, float, bool, str, type(None), tuple)) else "<object>"])
        return out

    surface = {
        "MobilePoserNet": {n: sig(getattr(MobilePoserNet, n)) for n in ("__init__", "from_pretrained", "reset", "_prob_to_weight", "_reduced_global_to_full", "forward", "forward_offline", "forward_online")},
        "Joints": {"forward": sig(Joints.forward)}, "Poser": {"forward": sig(Poser.forward), "_reduced_global_to_full": sig(Poser._reduced_global_to_full)},
        "FootContact": {"forward": sig(FootContact.forward)}, "Velocity": {"forward": sig(Velocity.forward), "forward_online": sig(Velocity.forward_online)},
        "ParametricModel": {n: sig(getattr(art.model.ParametricModel, n)) for n in ("__init__", "get_zero_pose_joint_and_vertex", "forward_kinematics_R", "inverse_kinematics_R", "forward_kinematics")},
        "PoseDataset": {n: sig(getattr(PoseDataset, n)) for n in ("__init__", "__len__", "__getitem__")},
        "PoseEvaluator": {n: sig(getattr(ref_eval.PoseEvaluator, n)) for n in ("__init__", "eval", "print")},
        "functions": {"load_model": sig(ref_load_model), "evaluate_pose": sig(ref_eval.evaluate_pose)},
    }
    m = new_model()
    surface["MobilePoserNet_attributes"] = sorted(k for k in vars(m) if not k.startswith("_") and k not in (
        "training", "hypers", "validation_step_loss", "training_step_loss"))          # (module internals / training bookkeeping aside)
    surface["MobilePoserNet_submodules"] = sorted(k for k, _ in m.named_children())
    with open(os.path.join(HERE, "g21_call_surface.json"), "w") as f:
        json.dump(surface, f, indent=0, sort_keys=True)

    print("golden vectors written to", HERE)
    for fn in sorted(os.listdir(HERE)):
        print("  %-24s %8d B" % (fn, os.path.getsize(os.path.join(HERE, fn))))


if __name__ == "__main__":
    main()

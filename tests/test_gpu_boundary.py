"""GPU tests of the drop-in boundary the way the reference's own callers reach it:

  a16 / f2   PoseDataset eval-mode formation with the ground-truth joints from the GPU forward kinematics, against the
             reference's own PoseDataset output (golden G8)                       (/root/reference/mobileposer/data.py:57-107)
  b          the reference's FILE layout, end to end, in a fresh process whose cwd holds
               smpl/basicmodel_m.pkl (latin1 pickle, scipy-sparse J_regressor)    (models/net.py:37, articulate/model.py:26-37)
               checkpoints/weights.pth (torch.save'd state dict)                  (utils/model_utils.py:6-15)
               data/processed_datasets/eval/dip_test.pt                           (config.py:33-34, data.py:45-55)
             and runs INTEGRATION.md section 1 literally: load_model(path), PoseDataset(fold='test', evaluate='dip'),
             forward_offline -- checked against goldens G8 and G4.
  a12        forward_kinematics(pose, shape=...) against golden G12 (articulate/model.py:84-89,208-240)
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO, cu, load_golden, npy

pytestmark = pytest.mark.gpu


def _g8_inputs(torch):
    g = load_golden("g8_dataset.npz")
    return g, {k: [torch.from_numpy(g[f"in{i}_{k}"]) for i in range(2)] for k in ("acc", "ori", "pose", "tran")}


def _check_g8(ds, g):
    assert len(ds) == int(g["n_items"]) == 24                      # 2 sequences x 12 combos
    for idx in range(len(ds)):
        imu, pose, joint, tran = ds[idx]
        assert np.array_equal(npy(imu), g[f"item{idx}_imu"]), idx
        assert np.array_equal(npy(pose), g[f"item{idx}_pose"]), idx
        assert np.array_equal(npy(tran), g[f"item{idx}_tran"]), idx
        assert np.abs(npy(joint) - g[f"item{idx}_joint"]).max() < 1e-5, idx


def test_g8_dataset_with_gpu_fk(torch_mod, net):
    """PoseDataset(..., fk=net.forward_kinematics): all 24 items of G8 (imu / pose / tran exact, joint <= 1e-5)."""
    from mobileposer_amd.data import PoseDataset
    g, data = _g8_inputs(torch_mod)
    _check_g8(PoseDataset(fold='test', evaluate='dip', data=data, fk=net.forward_kinematics), g)
    assert net.device_error() == 0


def test_g8_dataset_default_body_model_is_the_gpu(torch_mod, smpl):
    """Without a callable PoseDataset builds its own ParametricModel (data.py:24) -- a body-only native handle
    (mp_create_body); the network entry points refuse such a handle."""
    import ctypes as C
    from mobileposer_amd import _lib
    from mobileposer_amd.data import PoseDataset
    g, data = _g8_inputs(torch_mod)
    ds = PoseDataset(fold='test', evaluate='dip', data=data, smpl=smpl)
    _check_g8(ds, g)
    bm = ds.bodymodel
    assert bm is not None and bm._own is not None
    lib = _lib.load()
    x = torch_mod.zeros(1, 4, 60, device="cuda:0")
    o = [torch_mod.empty(4, 24, 3, 3, device="cuda:0"), torch_mod.empty(1, 4, 72, device="cuda:0"),
         torch_mod.empty(1, 4, 72, device="cuda:0"), torch_mod.empty(1, 4, 2, device="cuda:0")]
    rc = lib.mp_forward(bm._own, C.c_void_p(x.data_ptr()), (C.c_int32 * 1)(4), 1, 4, *[C.c_void_p(t.data_ptr()) for t in o],
                        None, None)
    assert rc == _lib.MP_ERR_INVALID and "body-only" in _lib.last_error(bm._own)
    # the stand-alone body model agrees with the reference's FK golden, mesh included
    g6 = load_golden("g6_fk.npz")
    Rg, jg, vg = bm.forward_kinematics(cu(torch_mod, g6["pose"]), tran=cu(torch_mod, g6["tran"]), calc_mesh=True)
    assert np.abs(npy(Rg) - g6["R_global"]).max() < 1e-5
    assert np.abs(npy(jg) - g6["joint_tran"]).max() < 1e-5 and np.abs(npy(vg) - g6["vert_tran"]).max() < 1e-5
    bm.close()


_LAYOUT_SCRIPT = r'''
import os, pickle, sys
import numpy as np, scipy.sparse, torch
repo, golden = sys.argv[1], sys.argv[2]
sys.path.insert(0, repo)
from mobileposer_amd import synthetic
# ---- the reference's file layout under the current directory (config.py:26-38) ----
smpl = synthetic.synthetic_smpl()
pk = dict(smpl)
pk["J_regressor"] = scipy.sparse.csc_matrix(smpl["J_regressor"])
os.makedirs("smpl"); os.makedirs("checkpoints"); os.makedirs("data/processed_datasets/eval")
with open("smpl/basicmodel_m.pkl", "wb") as f:
    pickle.dump(pk, f)
torch.save({k: torch.from_numpy(v) for k, v in synthetic.make_weights(0).items()}, "checkpoints/weights.pth")
g8 = dict(np.load(os.path.join(golden, "g8_dataset.npz")))
torch.save({k: [torch.from_numpy(g8["in%d_%s" % (i, k)]) for i in range(2)] for k in ("acc", "ori", "pose", "tran")},
           "data/processed_datasets/eval/dip_test.pt")
# ---- INTEGRATION.md section 1, literally: only the imports differ from the reference's evaluate.py:111-126 ----
from mobileposer_amd.config import paths
from mobileposer_amd.data import PoseDataset
from mobileposer_amd.model_utils import load_model
assert str(paths.smpl_file) == os.path.join(os.getcwd(), "smpl/basicmodel_m.pkl")
model = load_model(paths.weights_file)                       # utils/model_utils.py:6-15: the path and nothing else
dataset = PoseDataset(fold='test', evaluate='dip')           # evaluate.py:124
assert model.bodymodel.face is not None and model.n_vertex == smpl["v_template"].shape[0]      # came from the pickle
assert len(dataset) == 24
for idx in range(24):
    imu, pose, joint, tran = dataset[idx]
    assert np.array_equal(imu.numpy(), g8["item%d_imu" % idx]) and np.array_equal(pose.numpy(), g8["item%d_pose" % idx])
    assert np.array_equal(tran.numpy(), g8["item%d_tran" % idx])
    assert np.abs(joint.cpu().numpy() - g8["item%d_joint" % idx]).max() < 1e-5
# ---- golden G4 (forward_offline x3 with the stale-velocity-state quirk) through the file-loaded model ----
g4 = dict(np.load(os.path.join(golden, "g4_offline.npz")))
def geo(a, b):
    D = np.swapaxes(a.astype(np.float64), -1, -2) @ b.astype(np.float64)
    n = np.linalg.norm(D - np.eye(3), axis=(-1, -2))
    return 2.0 * np.arcsin(np.clip(n / (2.0 * np.sqrt(2.0)), 0.0, 1.0))
for tag, key in (("a", "imu_a"), ("b", "imu_b"), ("a_again", "imu_a")):
    model.reset()
    pose, joints, tran, contact = model.forward_offline(torch.from_numpy(g4[key]).cuda(), [g4[key].shape[1]])
    assert np.abs(joints.cpu().numpy() - g4[tag + "_joints"]).max() < 1e-4, tag
    assert np.abs(contact.cpu().numpy() - g4[tag + "_contact"]).max() < 1e-4, tag
    assert geo(pose.cpu().numpy(), g4[tag + "_pose"]).max() < 1e-4, tag
    assert np.abs(tran.cpu().numpy() - g4[tag + "_tran"]).max() < 1e-3, tag
assert model.device_error() == 0
# ---- golden G21: every attribute and sub-module the reference's instance has (net.py:28-74) is there ----
import json
g21 = json.load(open(os.path.join(golden, "g21_call_surface.json")))
for name in g21["MobilePoserNet_attributes"] + g21["MobilePoserNet_submodules"]:
    assert hasattr(model, name), name
# ---- evaluate.py:16-18: PoseEvaluator() with no argument builds its body model from paths.smpl_file ----
from mobileposer_amd.evaluate import PoseEvaluator
g9 = dict(np.load(os.path.join(golden, "g9_evaluator.npz")))
ev_file, ev_model = PoseEvaluator(), PoseEvaluator(model)
args = [torch.from_numpy(g9[k]).cuda() for k in ("pose_p", "pose_t")]
kw = {k: torch.from_numpy(g9[k]).cuda() for k in ("tran_p", "tran_t")}
t_file, t_model = ev_file.eval(*args, **kw), ev_model.eval(*args, **kw)
assert t_file.shape == (8, 2) and torch.equal(t_file, t_model)
ev_file.model.close()
model.close()
print("LAYOUT-OK")
'''


def test_reference_file_layout_end_to_end(tmp_path):
    env = dict(os.environ)
    env["PYTHONPATH"] = REPO + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", _LAYOUT_SCRIPT, REPO, GOLDEN], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "LAYOUT-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_g12_forward_kinematics_with_shape(torch_mod, net):
    """forward_kinematics(pose, shape, tran, calc_mesh) for one shared shape ([10]) and per-frame shapes ([N,10])."""
    g = load_golden("g12_fk_shape.npz")
    pose, tran = cu(torch_mod, g["pose"]), cu(torch_mod, g["tran"])
    for tag in ("one", "per"):
        shape = cu(torch_mod, g[f"{tag}_shape"])
        Rg, jg, vg = net.forward_kinematics(pose, tran=tran, calc_mesh=True, shape=shape)
        assert np.abs(npy(Rg) - g[f"{tag}_R"]).max() < 1e-5, tag
        assert np.abs(npy(jg) - g[f"{tag}_joint"]).max() < 1e-5, tag
        assert np.abs(npy(vg) - g[f"{tag}_vert"]).max() < 1e-5, tag
        Rg2, jg2 = net.bodymodel.forward_kinematics(pose, shape=shape)          # no mesh, no translation
        assert np.abs(npy(jg2) - g[f"{tag}_joint_notran"]).max() < 1e-5, tag
    with pytest.raises(RuntimeError):
        net.forward_kinematics(pose, shape=torch_mod.zeros(3, 10))
    assert net.device_error() == 0


def test_library_calls_leave_the_current_device_alone(torch_mod, weights, smpl):
    """Every entry point selects its handle's GPU for the duration of the call and puts the caller's current device back
    (torch reads hipGetDevice() as its own current device).  With one GPU this pins that calls -- including create /
    destroy, the state setters and the mode switches, which touch HIP outside enter() -- do not move it; with two or more
    GPUs the handle lives on the LAST one while device 0 stays current, and the G4 golden must still come out of it."""
    from mobileposer_amd.net import MobilePoserNet
    torch = torch_mod
    n_dev = torch.cuda.device_count()
    dev = "cuda:%d" % (n_dev - 1)
    torch.cuda.set_device(0)
    g = load_golden("g4_offline.npz")
    with MobilePoserNet.from_numpy(weights, smpl, device=dev) as n:
        assert torch.cuda.current_device() == 0
        n.reset()
        pose, joints, tran, contact = n.forward_offline(torch.from_numpy(g["imu_a"]).to(dev), [g["imu_a"].shape[1]])
        assert torch.cuda.current_device() == 0 and pose.device == torch.device(dev)
        assert np.abs(npy(tran) - g["a_tran"]).max() < 1e-3
        assert np.abs(npy(joints) - g["a_joints"]).max() < 1e-4
        n.velocity.rnn_state = n.velocity.rnn_state                  # mp_get / mp_set_velocity_state
        n.set_lstm_mode(0); n.set_lstm_mode(1); n.set_graph_mode(0)
        Rg, jg = n.forward_kinematics(pose)
        assert jg.device == torch.device(dev) and n.device_error() == 0
        n.finish()
        assert torch.cuda.current_device() == 0
    assert torch.cuda.current_device() == 0
